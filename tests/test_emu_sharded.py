"""CPU tier: the sharded-spectrum transforms of the C-ABI (sc_transform_forward_sharded / _inverse_sharded /
sc_bias_grad_sharded, include/sc_engine.h: the rank-major all-to-all buffer [block][image][rows][rest] written and
read in place by the transforms of a mode-parallel layer) in host emulation against the plain transforms plus the
permutation they replace.  Covers the fused 2-D kernels (native addressing in the store / load loops), the
size-agnostic passes in 2-d and 3-d (staging copy + one permutation launch), block counts that do and do not divide
the first kept dim (zero rows on the wire), both forward-type and both inverse-type modes, and the argument checks."""
import numpy as np
import pytest
import torch

from engine_runner import emu_lib
from neuraloperator_amd import _lib
from neuraloperator_amd.modes import halve_last_mode, kept_block


@pytest.fixture(scope="module")
def lib():
    return emu_lib()


def _to_shards(xh, P, rows):                      # (n, k1, rest.., 2) -> [P, n, rows, rest.., 2], zero rows past k1
    n, k1 = xh.shape[:2]
    pad = xh.new_zeros((n, P * rows, *xh.shape[2:]))
    pad[:, :k1] = xh
    return pad.unflatten(1, (P, rows)).movedim(1, 0).contiguous()


CASES = [
    # spatial, n_modes, P
    ((64, 256), (16, 12), 4),      # fused 2-D kernels, 16 rows / 4
    ((64, 256), (10, 64), 4),      # ... 10 rows over 4 blocks of 3: two zero rows on the wire
    ((128, 256), (64, 64), 8),     # ... the metric geometry's kept block over 8 ranks
    ((12, 10), (6, 6), 2),         # size-agnostic passes
    ((12, 10), (5, 6), 2),         # ... padded
    ((6, 8, 10), (4, 4, 4), 4),    # 3-d: one row per block
    ((9, 8, 10), (5, 4, 6), 3),    # 3-d padded, odd first dim
    ((64, 64, 64), (8, 8, 8), 4),  # round 5: k_ax64 addresses the shards natively (no staging buffer, no permutation launch)
    ((64, 64, 64), (7, 8, 8), 4),  # ... 7 rows over 4 blocks of 2: a zero row on the wire
    ((64, 256), (6, 12), 4),       # round 5: 6 rows over 4 blocks of 2 -- the last block is EMPTY (fused kernels, native)
    ((12, 10), (5, 6), 4),         # ... 5 rows over 4 blocks of 2: a half-empty and an empty block (permutation launch)
]


@pytest.mark.parametrize("case", CASES, ids=lambda c: "x".join(map(str, c[0])) + f"_P{c[2]}")
def test_sharded_transforms_match_plain_plus_permutation(lib, case):
    spatial, modes, P = case
    torch.manual_seed(3)
    nm = halve_last_mode(modes)
    kept, _ = kept_block(list(spatial), nm, nm)
    k1, rest = kept[0], int(np.prod(kept[1:]))
    rows = -(-k1 // P)
    n, c = (1, 2) if int(np.prod(spatial)) >= 64 ** 3 else (2, 3)      # (64^3 volumes in host emulation: CPU-tier time)
    ni = n * c
    plan = lib.plan_create(list(spatial), kept)
    x = torch.randn(n, c, *spatial)
    ws = torch.empty(lib.plan_workspace_bytes_sharded(plan, ni) + 256, dtype=torch.uint8)
    sh = lib.shards(P, rows, ni * rows * rest)
    for mode in (_lib.SC_FWD_SCALED, _lib.SC_FWD_ADJ_C2R):
        plain = torch.empty(ni, *kept, 2)
        lib.transform_forward(plan, mode, x.data_ptr(), plain.data_ptr(), ni, ws.data_ptr())
        buf = torch.full((P, ni, rows, *kept[1:], 2), float("nan"))
        lib.transform_forward_sharded(plan, mode, x.data_ptr(), buf.data_ptr(), ni, sh, ws.data_ptr())
        assert torch.equal(buf, _to_shards(plain, P, rows)), mode
    # bias gradient off the sharded adjoint spectrum
    gb0, gb1 = torch.empty(c), torch.empty(c)
    lib.bias_grad(plan, plain.data_ptr(), n, c, gb0.data_ptr())
    lib.bias_grad_sharded(plan, buf.data_ptr(), n, c, sh, gb1.data_ptr())
    assert torch.equal(gb0, gb1)
    # inverse-type transforms reading the sharded buffer in place
    yh = torch.randn(ni, *kept, 2)
    ybuf = _to_shards(yh, P, rows)
    if P * rows != k1:                             # whatever sits in the padding rows of the wire buffer is ignored
        ybuf.view(P, ni, rows, -1)[P - 1, :, k1 - (P - 1) * rows:] = 7.0
    bias = torch.randn(c)
    for mode, b in ((_lib.SC_INV_PADDED, bias), (_lib.SC_INV_ADJ_R2C, None)):
        y0 = torch.empty(n, c, *spatial)
        y1 = torch.full((n, c, *spatial), float("nan"))
        bp = 0 if b is None else b.data_ptr()
        lib.transform_inverse(plan, mode, yh.data_ptr(), bp, c, y0.data_ptr(), ni, ws.data_ptr())
        lib.transform_inverse_sharded(plan, mode, ybuf.data_ptr(), bp, c, y1.data_ptr(), ni, sh, ws.data_ptr())
        assert torch.equal(y0, y1), mode
    # a block stride larger than a block (slices of a bigger buffer)
    big = torch.full((P, 2 * ni, rows, *kept[1:], 2), float("nan"))
    sh2 = lib.shards(P, rows, 2 * ni * rows * rest)
    lib.transform_forward_sharded(plan, _lib.SC_FWD_ADJ_C2R, x.data_ptr(), big.data_ptr(), ni, sh2, ws.data_ptr())
    assert torch.equal(big[:, :ni], buf)
    if (P - 1) * rows < k1:
        with pytest.raises(RuntimeError):          # blocks must cover the first kept dim
            lib.transform_forward_sharded(plan, _lib.SC_FWD_SCALED, x.data_ptr(), buf.data_ptr(), ni,
                                          lib.shards(P - 1, rows, ni * rows * rest), ws.data_ptr())
    with pytest.raises(RuntimeError):              # block stride smaller than a block
        lib.transform_forward_sharded(plan, _lib.SC_FWD_SCALED, x.data_ptr(), buf.data_ptr(), ni,
                                      lib.shards(P, rows, ni * rows * rest - 1), ws.data_ptr())
    lib.plan_destroy(plan)
