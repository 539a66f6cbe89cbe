"""CPU tier: k_fft2d_fwd_mx (round 5: the row pass of the forward-type transform of bfloat16 tensors on the matrix
cores, neuraloperator_amd/csrc/sc_kernels_fft3mx.h) in host emulation, at transform level: several images per persistent
workgroup (the emulated chip has one compute unit: the grid is two workgroups), so that the two-groups-ahead row
requests cross image boundaries and -- at H = 64, one group per image -- the two register sets alternate over images
with an odd image left over; all three heights, both forward-type modes, kept blocks smaller than 64 x 33 and the
sharded spectrum layout.  Against a float64 rfft2 of the same bf16 values and against the vector-ALU kernel
(SC_PLAN_NO_MX_FFT)."""
import numpy as np
import pytest
import torch

from engine_runner import emu_lib
from neuraloperator_amd import _lib


@pytest.fixture(scope="module")
def lib():
    return emu_lib()


def _ref(x, H, Mx, My, mode):
    # mode 0: rfft2(norm="forward") on the kept block; mode 1: adjoint of the zero-padded C2R with norm="forward"
    # (no 1 / (H W); interior columns count twice)
    X = np.fft.rfft2(x.astype(np.float64), axes=(-2, -1))
    rows = np.r_[np.arange(H - Mx // 2, H), np.arange(0, Mx - Mx // 2)] if Mx > 1 else np.array([0])
    K = X[..., rows, :][..., :My]
    if mode == _lib.SC_FWD_SCALED:
        return K / (H * 256)
    s = np.full(My, 2.0)
    s[0] = 1.0
    return K * s


@pytest.mark.parametrize("H,Mx,My,n_img", [(256, 64, 33, 3), (256, 20, 9, 3), (128, 64, 33, 5), (64, 64, 33, 7),
                                            (64, 12, 33, 4)])
def test_mx_forward_vs_float64_and_valu(lib, H, Mx, My, n_img):
    torch.manual_seed(H + Mx + n_img)
    x = torch.randn(n_img, H, 256).bfloat16()
    xf = x.float().numpy()
    for mode in (_lib.SC_FWD_SCALED, _lib.SC_FWD_ADJ_C2R):
        got = {}
        # (the vector-ALU kernel beside it on the small cases only: CPU-tier time)
        routes = (("mx", _lib.SC_PLAN_IO_BF16), ("valu", _lib.SC_PLAN_IO_BF16 | _lib.SC_PLAN_NO_MX_FFT))
        for tag, fl in routes[:2 if H * n_img <= 512 else 1]:
            plan = lib.plan_create([H, 256], [Mx, My], flags=fl)
            assert lib.plan_kernel_name(plan, 0) == ("k_fft2d_fwd_mx" if tag == "mx" else "k_fft2d_fwd3")
            xh = torch.full((n_img, Mx, My, 2), float("nan"))
            lib.transform_forward(plan, mode, x.data_ptr(), xh.data_ptr(), n_img, 0)
            got[tag] = torch.view_as_complex(xh).numpy()
            lib.plan_destroy(plan)
        ref = _ref(xf, H, Mx, My, mode)
        for tag, g in got.items():
            e = np.linalg.norm(g - ref) / np.linalg.norm(ref)
            assert e < 1e-6, (tag, mode, e)


def test_mx_forward_sharded_layout(lib):
    H, Mx, My, n_img, P = 128, 64, 33, 3, 8
    rows = Mx // P
    torch.manual_seed(9)
    x = torch.randn(n_img, H, 256).bfloat16()
    plan = lib.plan_create([H, 256], [Mx, My], flags=_lib.SC_PLAN_IO_BF16)
    assert lib.plan_kernel_name(plan, 0) == "k_fft2d_fwd_mx"
    plain = torch.empty(n_img, Mx, My, 2)
    lib.transform_forward(plan, _lib.SC_FWD_SCALED, x.data_ptr(), plain.data_ptr(), n_img, 0)
    buf = torch.full((P, n_img, rows, My, 2), float("nan"))
    sh = lib.shards(P, rows, n_img * rows * My)
    lib.transform_forward_sharded(plan, _lib.SC_FWD_SCALED, x.data_ptr(), buf.data_ptr(), n_img, sh, 0)
    want = plain.unflatten(1, (P, rows)).movedim(1, 0).contiguous()
    assert torch.equal(buf, want)
    lib.plan_destroy(plan)
