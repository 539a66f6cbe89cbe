"""CPU tier: the factorised last-two-axes kernels for 64 x 64 planes (sc_kernels_plane64.h) in host emulation, all
four transform modes through the C-ABI, 2-D and 3-D plans, against numpy's FFT of the same definition (as
test_emu_plane128.py; spectral_convolution.py:443-449, 500-519, 531-568 and their adjoints)."""
import numpy as np
import pytest
import torch

from neuraloperator_amd import _lib
from engine_runner import emu_lib, rel_l2
from test_emu_plane128 import _ref_forward, _ref_inverse

TOL = 2e-6


@pytest.fixture(scope="module")
def lib():
    return emu_lib()


CASES = [
    ((64, 64), (32, 17), 3),            # 2-D grid, the largest kept block (n_modes = 32)
    ((64, 64), (16, 9), 2),             # n_modes = 16
    ((64, 64), (21, 12), 2),            # odd kept rows
    ((64, 64), (1, 1), 1),
    ((5, 64, 64), (4, 32, 17), 2),      # 3-D: planes + the size-agnostic axis pass over the first dim
    ((64, 64, 64), (16, 16, 9), 1),     # FNO3d 64^3, n_modes = 16: first axis on k_ax64
    ((64, 64, 64), (27, 32, 17), 1),    # odd kept count on the first axis
]


@pytest.mark.parametrize("spatial,kept,n_img", CASES, ids=["x".join(map(str, s)) + "_k" + "x".join(map(str, k)) for s, k, _ in CASES])
@pytest.mark.parametrize("norm", ["forward", "ortho"])
def test_plane64_kernels(lib, spatial, kept, n_img, norm):
    if norm == "ortho" and len(spatial) == 3:
        pytest.skip("one norm is enough for the 3-D cases")
    rng = np.random.default_rng(11)
    plan = lib.plan_create(list(spatial), list(kept), fft_norm=norm, flags=0)
    ntot = int(np.prod(spatial))
    sf, si = (1.0 / ntot, 1.0) if norm == "forward" else (ntot ** -0.5, ntot ** -0.5)
    try:
        assert lib.plan_kernel_name(plan, 0) == "k_pl64_fwd" and lib.plan_kernel_name(plan, 1) == "k_pl64_inv"
        ws = torch.empty(max(lib.plan_workspace_bytes(plan, n_img), 256), dtype=torch.uint8)
        x = torch.from_numpy(rng.standard_normal((n_img, *spatial)).astype(np.float32))
        for mode, scale, weighted in ((_lib.SC_FWD_SCALED, sf, False), (_lib.SC_FWD_ADJ_C2R, si, True)):
            xhat = torch.full((n_img, *kept), float("nan"), dtype=torch.complex64)
            lib.transform_forward(plan, mode, x.data_ptr(), torch.view_as_real(xhat).data_ptr(), n_img, ws.data_ptr(), 0)
            assert rel_l2(xhat.numpy(), _ref_forward(x.numpy(), kept, scale, weighted)) < TOL, f"forward mode {mode}"
        yh = (rng.standard_normal((n_img, *kept)) + 1j * rng.standard_normal((n_img, *kept))).astype(np.complex64)
        yhat = torch.from_numpy(yh)
        bias = torch.from_numpy(rng.standard_normal(n_img).astype(np.float32))
        for mode, scale, weighted, b in ((_lib.SC_INV_PADDED, si, True, bias), (_lib.SC_INV_ADJ_R2C, sf, False, None)):
            y = torch.full((n_img, *spatial), float("nan"), dtype=torch.float32)
            lib.transform_inverse(plan, mode, torch.view_as_real(yhat).data_ptr(), 0 if b is None else b.data_ptr(),
                                  n_img, y.data_ptr(), n_img, ws.data_ptr(), 0)
            ref = _ref_inverse(yh, spatial, scale, weighted)
            if b is not None:
                ref = ref + b.numpy().astype(np.float64).reshape((n_img,) + (1,) * len(spatial))
            assert rel_l2(y.numpy(), ref) < TOL, f"inverse mode {mode}"
    finally:
        lib.plan_destroy(plan)


def test_plane64_scope(lib):
    """Larger kept blocks (the Nyquist column, more than 32 rows) and mixed plane sizes stay on the other kernels."""
    for spatial, kept in (((64, 64), (64, 17)), ((64, 64), (32, 33)), ((64, 128), (32, 17)), ((32, 64), (16, 17))):
        plan = lib.plan_create(list(spatial), list(kept), fft_norm="forward", flags=0)
        try:
            assert lib.plan_kernel_name(plan, 0) != "k_pl64_fwd"
        finally:
            lib.plan_destroy(plan)


@pytest.mark.parametrize("spatial,kept,name", [((64, 64), (32, 17), "k_pl64"), ((128, 128), (32, 17), "k_pl128"),
                                               ((3, 64, 64), (2, 16, 9), "k_pl64")], ids=["pl64", "pl128", "pl64_3d"])
def test_misaligned_planes_take_the_size_agnostic_passes(lib, spatial, kept, name):
    """the one-workgroup plane kernels move rows with 16-byte accesses.  A contiguous view with an odd storage
    offset (x = flat[1:1 + n].view(...), a gradient handed over out of a bucketed buffer) used to make the 64 x 64 route
    FAIL the call; the dispatch now falls back to the size-agnostic passes of the same plan -- same results to rounding."""
    rng = np.random.default_rng(5)
    n_img = 2
    n = n_img * int(np.prod(spatial))
    plan = lib.plan_create(list(spatial), list(kept), fft_norm="forward", flags=0)
    try:
        assert lib.plan_kernel_name(plan, 0).startswith(name)
        ws = torch.empty(max(lib.plan_workspace_bytes(plan, n_img), 256), dtype=torch.uint8)
        base = torch.from_numpy(rng.standard_normal(n + 4).astype(np.float32))
        assert base.data_ptr() % 16 == 0
        want = None
        x0 = base[:n].clone().view(n_img, *spatial)
        for off in (0, 1, 3):
            buf = torch.zeros(n + 4)
            x = buf[off:off + n].view(n_img, *spatial)
            x.copy_(x0)
            assert x.data_ptr() % 16 == (4 * off) % 16
            xhat = torch.full((n_img, *kept), float("nan"), dtype=torch.complex64)
            lib.transform_forward(plan, _lib.SC_FWD_SCALED, x.data_ptr(), torch.view_as_real(xhat).data_ptr(), n_img,
                                  ws.data_ptr(), 0)
            if want is None:
                want = xhat.clone()
                assert rel_l2(xhat.numpy(), _ref_forward(x0.numpy(), kept, 1.0 / int(np.prod(spatial)), False)) < TOL
            else:
                assert rel_l2(xhat.numpy(), want.numpy()) < TOL
        yh = (rng.standard_normal((n_img, *kept)) + 1j * rng.standard_normal((n_img, *kept))).astype(np.complex64)
        yhat = torch.from_numpy(yh)
        ref = _ref_inverse(yh, spatial, 1.0, True)
        for off in (0, 2):
            buf = torch.full((n + 4,), float("nan"))
            y = buf[off:off + n].view(n_img, *spatial)
            lib.transform_inverse(plan, _lib.SC_INV_PADDED, torch.view_as_real(yhat).data_ptr(), 0, n_img, y.data_ptr(), n_img,
                                  ws.data_ptr(), 0)
            assert rel_l2(y.numpy(), ref) < TOL
    finally:
        lib.plan_destroy(plan)
