"""GPU tier: the 64 x 64 plane kernels (sc_kernels_plane64.h) on the device, every transform mode through the C-ABI
against numpy's FFT of the same definition -- the shapes of the CPU-tier test (test_emu_plane64.py) plus batches that
fill the chip several times over (wave-local LDS aliasing and the staged 16-byte row accesses only run for real here)."""
import numpy as np
import pytest
import torch

from neuraloperator_amd import _lib
from engine_runner import rel_l2
from test_emu_plane128 import _ref_forward, _ref_inverse

pytestmark = pytest.mark.gpu
TOL = 2e-6

CASES = [
    ((64, 64), (32, 17), 2100),         # more planes than 4 workgroups x 256 units x 2
    ((64, 64), (16, 9), 37),
    ((64, 64), (21, 12), 5),            # odd kept rows
    ((64, 64), (1, 1), 3),
    ((5, 64, 64), (4, 32, 17), 9),      # first axis on the size-agnostic pass
    ((64, 64, 64), (16, 16, 9), 6),     # FNO3d 64^3: k_ax64
    ((64, 64, 64), (27, 32, 17), 2),
]


@pytest.fixture(scope="module")
def lib():
    if not torch.cuda.is_available():
        pytest.skip("GPU tier: no GPU visible")
    return _lib.get_lib()


@pytest.mark.parametrize("spatial,kept,n_img", CASES, ids=["x".join(map(str, s)) + "_k" + "x".join(map(str, k)) for s, k, _ in CASES])
def test_plane64_kernels_on_device(lib, spatial, kept, n_img):
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(13)
    plan = lib.plan_create(list(spatial), list(kept), fft_norm="forward", flags=0)
    ntot = int(np.prod(spatial))
    sf, si = 1.0 / ntot, 1.0
    try:
        assert lib.plan_kernel_name(plan, 0) == "k_pl64_fwd" and lib.plan_kernel_name(plan, 1) == "k_pl64_inv"
        ws = torch.empty(max(lib.plan_workspace_bytes(plan, n_img), 256), dtype=torch.uint8, device=dev)
        xh = rng.standard_normal((n_img, *spatial)).astype(np.float32)
        x = torch.from_numpy(xh).to(dev)
        st = torch.cuda.current_stream().cuda_stream
        for mode, scale, weighted in ((_lib.SC_FWD_SCALED, sf, False), (_lib.SC_FWD_ADJ_C2R, si, True)):
            xhat = torch.full((n_img, *kept), float("nan"), dtype=torch.complex64, device=dev)
            lib.transform_forward(plan, mode, x.data_ptr(), torch.view_as_real(xhat).data_ptr(), n_img, ws.data_ptr(), st)
            assert rel_l2(xhat.cpu().numpy(), _ref_forward(xh, kept, scale, weighted)) < TOL, f"forward mode {mode}"
        yh = (rng.standard_normal((n_img, *kept)) + 1j * rng.standard_normal((n_img, *kept))).astype(np.complex64)
        yhat = torch.from_numpy(yh).to(dev)
        bh = rng.standard_normal(n_img).astype(np.float32)
        bias = torch.from_numpy(bh).to(dev)
        for mode, scale, weighted, b in ((_lib.SC_INV_PADDED, si, True, bias), (_lib.SC_INV_ADJ_R2C, sf, False, None)):
            y = torch.full((n_img, *spatial), float("nan"), dtype=torch.float32, device=dev)
            lib.transform_inverse(plan, mode, torch.view_as_real(yhat).data_ptr(), 0 if b is None else b.data_ptr(),
                                  n_img, y.data_ptr(), n_img, ws.data_ptr(), st)
            ref = _ref_inverse(yh, spatial, scale, weighted)
            if b is not None:
                ref = ref + bh.astype(np.float64).reshape((n_img,) + (1,) * len(spatial))
            assert rel_l2(y.cpu().numpy(), ref) < TOL, f"inverse mode {mode}"
    finally:
        lib.plan_destroy(plan)
