"""CPU tier: the LDS-DMA streamed matrix-core mode GEMM (sc_kernels_gemm8.h, k_modegemm_dma) in host emulation --
MFMA replaced by its documented lane / register map, LDS-DMA by a synchronous copy; unit decoding, the source-side
bank swizzle, ring rotation, operand sign masks, the patch epilogue and the clamping of ragged rows / columns / r
are the product source -- against a numpy complex128 einsum.  Covers the layer's three contractions (forward, gX
with conj(B) and transposed strides, gW with conj(A)), odd / short / long r loops (shorter and longer than the
ring), several row / column blocks per workgroup and per launch, hidden = 128."""
import numpy as np
import pytest
import torch

from engine_runner import emu_lib, rel_l2
from neuraloperator_amd import _lib

TOL = 1e-5


@pytest.fixture(scope="module")
def lib():
    return emu_lib()


def _rand(*shape, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.complex(torch.randn(*shape, generator=g), torch.randn(*shape, generator=g))


def run(lib, a, b, c, flags=0, expect=2, **kw):
    flags |= _lib.SC_GEMM_NO_SB            # this file pins the streamed / generation-1 kernels (small extents: test_emu_sb.py)
    assert lib.modegemm_path(flags=flags, **kw) == expect, "test must exercise the intended kernel"
    lib.modegemm(torch.view_as_real(a).data_ptr(), torch.view_as_real(b).data_ptr(),
                 torch.view_as_real(c).data_ptr(), 0, flags=flags, **kw)
    return c


# (B, Ci, Co, M, tiles per workgroup (0 = auto))
FWD = [
    (32, 64, 64, 16, 0),     # the metric tile: two mode groups, 32 stages (ring wraps five times)
    (32, 6, 64, 8, 0),       # r loop shorter than the ring (3 stages)
    (32, 7, 64, 8, 0),       # odd r: the clamped duplicate of the last pair contributes nothing
    (28, 20, 52, 8, 0),      # ragged rows and columns inside one tile
    (64, 16, 128, 8, 0),     # 2 row blocks x 2 column blocks, one tile per workgroup
    (64, 16, 128, 8, 4),     # ... all four tiles in one workgroup, back to back
    (56, 12, 64, 24, 2),     # second row block ragged (24 of 32 rows), two tiles per workgroup, 3 mode groups
    # mode counts that are multiples of 16: 128-byte segments, 8 waves, 32 x 32 tiles
    (32, 64, 64, 32, 0),     # the metric tile: two mode groups x two column blocks
    (28, 7, 52, 16, 0),      # ragged rows / columns (second column block 20 of 32), odd r
    (64, 16, 128, 16, 0),    # 2 row blocks x 4 column blocks
    (64, 16, 128, 16, 8),    # ... all eight tiles in one workgroup
    (56, 12, 64, 48, 2),     # ragged second row block, two tiles per workgroup, 3 mode groups
]


@pytest.mark.parametrize("case", FWD, ids=lambda c: "B%d_Ci%d_Co%d_M%d_t%d" % c)
def test_forward(lib, case):
    B, Ci, Co, M, bpw = case
    x, w = _rand(B, Ci, M, seed=1), _rand(Ci, Co, M, seed=2)
    y = torch.full((B, Co, M), float("nan"), dtype=torch.complex64)
    kw = dict(P=B, Q=Co, R=Ci, n_modes=M, a_sp=Ci * M, a_sr=M, a_sm=1, b_sr=Co * M, b_sq=M, b_sm=1,
              c_sp=Co * M, c_sq=M, c_sm=1)
    run(lib, x, w, y, flags=_lib.SC_GEMM_GRID(bpw), **kw)
    ref = np.einsum("bim,iom->bom", x.numpy().astype(np.complex128), w.numpy().astype(np.complex128))
    assert rel_l2(y.numpy(), ref) < TOL
    # and the generation-1 kernels on the same call agree
    y1 = torch.zeros_like(y)
    flags = _lib.SC_GEMM_NO_STREAM
    run(lib, x, w, y1, flags=flags, expect=lib.modegemm_path(flags=flags, **kw), **kw)
    assert rel_l2(y1.numpy(), ref) < TOL


@pytest.mark.parametrize("dims", [(32, 64, 64, 16), (32, 128, 128, 8), (30, 48, 70, 8), (32, 128, 128, 16), (30, 48, 70, 32),
                                  (8, 32, 32, 8), (11, 32, 40, 16)])
def test_gx_conj_b_transposed(lib, dims):
    """gxhat[b,i,m] = sum_o ghat[b,o,m] conj(W[i,o,m]): B operand read through transposed strides; the last two: a small
    batch (8 / 11 rows of one 32-row tile, the rest clamped duplicates that are never stored)"""
    B, Ci, Co, M = dims
    g, w = _rand(B, Co, M, seed=3), _rand(Ci, Co, M, seed=4)
    gx = torch.full((B, Ci, M), float("nan"), dtype=torch.complex64)
    run(lib, g, w, gx, P=B, Q=Ci, R=Co, n_modes=M, a_sp=Co * M, a_sr=M, a_sm=1, b_sr=M, b_sq=Co * M, b_sm=1,
        conj_b=1, c_sp=Ci * M, c_sq=M, c_sm=1)
    ref = np.einsum("bom,iom->bim", g.numpy().astype(np.complex128), np.conj(w.numpy().astype(np.complex128)))
    assert rel_l2(gx.numpy(), ref) < TOL


@pytest.mark.parametrize("dims", [(32, 64, 64, 16), (4, 128, 128, 8), (9, 56, 64, 8), (4, 128, 128, 16), (9, 56, 64, 32)])
def test_gw_conj_a(lib, dims):
    """gW[i,o,m] = sum_b conj(xhat[b,i,m]) ghat[b,o,m]: P = Ci (two row blocks at 64 -> both in one workgroup at
    the metric shape's mode count), streaming stores"""
    B, Ci, Co, M = dims
    x, g = _rand(B, Ci, M, seed=5), _rand(B, Co, M, seed=6)
    gw = torch.full((Ci, Co, M), float("nan"), dtype=torch.complex64)
    run(lib, x, g, gw, flags=_lib.SC_GEMM_STREAM_C, P=Ci, Q=Co, R=B, n_modes=M, a_sp=M, a_sr=Ci * M, a_sm=1,
        conj_a=1, b_sr=Co * M, b_sq=M, b_sm=1, c_sp=Co * M, c_sq=M, c_sm=1)
    ref = np.einsum("bim,bom->iom", np.conj(x.numpy().astype(np.complex128)), g.numpy().astype(np.complex128))
    assert rel_l2(gw.numpy(), ref) < TOL


def test_both_conjugates_and_padded_rows(lib):
    """conj(A) conj(B) (no layer call uses it, the C-ABI offers it); operands embedded in wider rows (strides larger
    than the extents, as a mode-sharded or padded caller would pass)"""
    P, Q, R, M, MS = 32, 64, 10, 8, 24
    a, b = _rand(P, R, MS, seed=7), _rand(R, Q, MS, seed=8)
    c = torch.zeros(P, Q, MS, dtype=torch.complex64)
    off = 8                                  # modes 8..15 of rows that hold 24
    av, bv, cv = (torch.view_as_real(t).reshape(-1)[2 * off:] for t in (a, b, c))
    kw = dict(P=P, Q=Q, R=R, n_modes=M, a_sp=R * MS, a_sr=MS, a_sm=1, b_sr=Q * MS, b_sq=MS, b_sm=1,
              c_sp=Q * MS, c_sq=MS, c_sm=1, conj_a=1, conj_b=1)
    assert lib.modegemm_path(**kw) == 2
    lib.modegemm(av.data_ptr(), bv.data_ptr(), cv.data_ptr(), 0, **kw)
    ref = np.einsum("prm,rqm->pqm", np.conj(a.numpy()[:, :, off:off + M].astype(np.complex128)),
                    np.conj(b.numpy()[:, :, off:off + M].astype(np.complex128)))
    assert rel_l2(c.numpy()[:, :, off:off + M], ref) < TOL
    assert not c.numpy()[:, :, :off].any() and not c.numpy()[:, :, off + M:].any()      # nothing outside the block


def test_eligibility(lib):
    base = dict(P=32, Q=64, R=64, n_modes=2112, a_sp=64 * 2112, a_sr=2112, a_sm=1, b_sr=64 * 2112, b_sq=2112,
                b_sm=1, c_sp=64 * 2112, c_sq=2112, c_sm=1)
    assert lib.modegemm_path(**base) == 2
    assert lib.modegemm_path(**dict(base, flags=_lib.SC_GEMM_NO_STREAM)) == 1
    assert lib.modegemm_path(**dict(base, flags=_lib.SC_GEMM_FORCE_VALU)) == 0
    assert lib.modegemm_path(**dict(base, n_modes=2110)) == 1            # not a multiple of 8
    assert lib.modegemm_path(**dict(base, a_sr=2111)) == 1               # rows not 16-byte aligned
    assert lib.modegemm_path(**dict(base, accumulate=1)) == 0
    # a small batch streams only against a weight read across its rows (gX-hat), and not below 8 rows
    gx = dict(base, P=8, b_sr=2112, b_sq=64 * 2112, conj_b=1)
    assert lib.modegemm_path(**gx) == 2
    assert lib.modegemm_path(**dict(gx, P=4)) == 3                        # round 3: <= 4 rows -> the small-extent streaming kernel
    assert lib.modegemm_path(**dict(gx, P=4, flags=_lib.SC_GEMM_NO_SB)) == 0
    assert lib.modegemm_path(**dict(base, P=8)) == 0                      # forward product: lanes-are-modes kernel
    assert lib.modegemm_path(**dict(base, Q=36)) == 2                    # ragged Tucker rank: round 3, tiles filled >= 1/2 stream
    assert lib.modegemm_path(**dict(base, Q=15)) != 2                    # 15 of 32 columns: below half a tile
    assert lib.modegemm_path(**dict(base, P=4, Q=128)) == 3              # 4 rows: neither matrix-core kernel
    assert lib.modegemm_path(**dict(base, P=4, Q=128, flags=_lib.SC_GEMM_NO_SB)) == 0
    assert lib.modegemm_path(**dict(base, P=128, Q=128, R=4)) == 3       # hidden 128 weight gradient at B = 4
    assert lib.modegemm_path(**dict(base, P=128, Q=128, R=4, flags=_lib.SC_GEMM_NO_SB)) == 2


@pytest.mark.parametrize("layout", ["AC", "B", "ABC"])
def test_tiled_operands(lib, layout):
    """mode-group-major ("tiled spectrum") operands: [group][row][col][16 modes] through a_sg / b_sg / c_sg"""
    B, Ci, Co, M = 32, 12, 64, 48
    G = M // 16
    x, w = _rand(B, Ci, M, seed=11), _rand(Ci, Co, M, seed=12)

    def tile(t):                                  # (P, Q, M) -> (G, P, Q, 16) contiguous
        p, q, _ = t.shape
        return t.reshape(p, q, G, 16).permute(2, 0, 1, 3).contiguous()

    def untile(t, p, q):
        return t.reshape(G, p, q, 16).permute(1, 2, 0, 3).reshape(p, q, M)

    a = tile(x) if "A" in layout else x
    b = tile(w) if "B" in layout else w
    c = torch.full((B, Co, M), float("nan"), dtype=torch.complex64)
    if "C" in layout:
        c = tile(c)
    kw = dict(P=B, Q=Co, R=Ci, n_modes=M, a_sm=1, b_sm=1, c_sm=1)
    kw.update(dict(a_sg=B * Ci * 16, a_sp=Ci * 16, a_sr=16) if "A" in layout else dict(a_sp=Ci * M, a_sr=M))
    kw.update(dict(b_sg=Ci * Co * 16, b_sr=Co * 16, b_sq=16) if "B" in layout else dict(b_sr=Co * M, b_sq=M))
    kw.update(dict(c_sg=B * Co * 16, c_sp=Co * 16, c_sq=16) if "C" in layout else dict(c_sp=Co * M, c_sq=M))
    run(lib, a, b, c, **kw)
    got = untile(c, B, Co) if "C" in layout else c
    ref = np.einsum("bim,iom->bom", x.numpy().astype(np.complex128), w.numpy().astype(np.complex128))
    assert rel_l2(got.numpy(), ref) < TOL
    # tiled operands on a call the streamed kernel does not take fail loudly (no silent mis-addressing)
    with pytest.raises(_lib.EngineError):
        lib.modegemm(torch.view_as_real(a).data_ptr(), torch.view_as_real(b).data_ptr(),
                     torch.view_as_real(c).data_ptr(), 0, **dict(kw, a_sg=16, flags=_lib.SC_GEMM_NO_STREAM))


# ---- the two contractions of a backward pass in one launch (k_modegemm_dma_bwd, sc_modegemm_pair) ----------------
def _pair_kw(B, Ci, Co, M):
    kw_w = dict(P=Ci, Q=Co, R=B, n_modes=M, a_sp=M, a_sr=Ci * M, a_sm=1, conj_a=1, b_sr=Co * M, b_sq=M, b_sm=1,
                c_sp=Co * M, c_sq=M, c_sm=1, flags=_lib.SC_GEMM_STREAM_C)
    kw_x = dict(P=B, Q=Ci, R=Co, n_modes=M, a_sp=Co * M, a_sr=M, a_sm=1, b_sr=M, b_sq=Co * M, b_sm=1, conj_b=1,
                c_sp=Ci * M, c_sq=M, c_sm=1)
    return kw_w, kw_x


# (B, Ci, Co, M, one launch?)
PAIR = [
    (32, 32, 32, 64, True),      # 8 + 8 workgroups: octets alternate between the jobs
    (32, 32, 64, 64, True),      # 16 (weight gradient) + 8: the longer job's trailing octet
    (64, 32, 32, 64, True),      # 8 + 16: ... the other way round
    (30, 48, 56, 32, True),      # ragged rows / columns in both jobs (16 + 8 workgroups)
    (8, 32, 32, 64, True),       # small batch: gX-hat with 8 real rows of its 32-row tile, r loop of 2 stages in gW
    (32, 32, 32, 8, False),      # 1 + 1 workgroups: not whole octets -> two launches, same results
    (32, 32, 32, 20, False),     # mode count not a multiple of 8 -> generation 1, two launches
]


@pytest.mark.parametrize("case", PAIR, ids=lambda c: "B%d_Ci%d_Co%d_M%d_%s" % (c[:4] + ("one" if c[4] else "two",)))
def test_backward_pair(lib, case):
    B, Ci, Co, M, fused = case
    x, g, w = _rand(B, Ci, M, seed=11), _rand(B, Co, M, seed=12), _rand(Ci, Co, M, seed=13)
    kw_w, kw_x = _pair_kw(B, Ci, Co, M)
    assert lib.modegemm_pair_fused(kw_w, kw_x) == fused, "test must exercise the intended launch"
    gw = torch.full((Ci, Co, M), float("nan"), dtype=torch.complex64)
    gx = torch.full((B, Ci, M), float("nan"), dtype=torch.complex64)
    p = lambda t: torch.view_as_real(t).data_ptr()
    lib.modegemm_pair(kw_w, p(x), p(g), p(gw), kw_x, p(g), p(w), p(gx))
    # the two single launches give the same BITS (same tiles, same k order)
    gw1, gx1 = torch.zeros_like(gw), torch.zeros_like(gx)
    lib.modegemm(p(x), p(g), p(gw1), 0, **kw_w)
    lib.modegemm(p(g), p(w), p(gx1), 0, **kw_x)
    assert torch.equal(torch.view_as_real(gw), torch.view_as_real(gw1))
    assert torch.equal(torch.view_as_real(gx), torch.view_as_real(gx1))
    x128, g128, w128 = (t.numpy().astype(np.complex128) for t in (x, g, w))
    assert rel_l2(gw.numpy(), np.einsum("bim,bom->iom", np.conj(x128), g128)) < TOL
    assert rel_l2(gx.numpy(), np.einsum("bom,iom->bim", g128, np.conj(w128))) < TOL


def test_pair_needs_the_backward_conjugations(lib):
    kw_w, kw_x = _pair_kw(32, 32, 32, 64)
    assert lib.modegemm_pair_fused(kw_w, kw_x)
    assert not lib.modegemm_pair_fused(kw_x, kw_w)                       # roles swapped
    assert not lib.modegemm_pair_fused(dict(kw_w, conj_a=0), kw_x)
    assert not lib.modegemm_pair_fused(kw_w, dict(kw_x, flags=_lib.SC_GEMM_NO_STREAM))
    assert not lib.modegemm_pair_fused(kw_w, dict(kw_x, n_modes=8))


def test_layer_backward_takes_the_pair_launch(lib):
    """sc_layer_backward at a shape whose two contractions qualify (kept block 8 x 8 = 64 modes, 32 x 32 channel
    tiles, batch 8: the small-batch rule for gX-hat): gW, gX-hat AND the bias gradient come out of
    k_modegemm_dma_bwd; against the oracle"""
    from engine_runner import layer_fwd_bwd
    from oracle import spectral_oracle as so
    B, C, nm, spatial = 8, 32, [8, 8], (16, 16)       # n_modes attribute (last entry already halved: 14 // 2 + 1)
    kw_w, kw_x = _pair_kw(B, C, C, 64)
    assert lib.modegemm_pair_fused(kw_w, kw_x)
    g0 = torch.Generator().manual_seed(21)
    x = torch.randn(B, C, *spatial, generator=g0)
    gy = torch.randn(B, C, *spatial, generator=g0)
    w = _rand(C, C, 8, 8, seed=22) * 0.1
    bias = torch.randn(C, 1, 1, generator=g0)
    y, gx, gw, gb, _ = layer_fwd_bwd(lib, x, w, bias, gy, nm, nm)
    xr, wr, br = x.clone().requires_grad_(True), w.clone().requires_grad_(True), bias.clone().requires_grad_(True)
    yo = so.forward_torch(xr, wr, br, nm, nm)
    yo.backward(gy)
    assert rel_l2(y.numpy(), yo.detach().numpy()) < TOL
    assert rel_l2(gx.numpy(), xr.grad.numpy()) < TOL
    assert rel_l2(gw.numpy(), wr.grad.numpy()) < TOL
    assert rel_l2(gb.numpy(), br.grad.numpy()) < TOL
