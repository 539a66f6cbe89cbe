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
        # mx = the default of bf16-I/O plans since round 6: TWO bf16 terms per twiddle (1.3e-6: 3000 x below the 2^-9 of the bf16
        # input, 8 x below the 1e-5 bar of the fp32 gradients it feeds); mx3 = SC_PLAN_MX_FFT_3TERM: fp32 round-off class
        routes = (("mx", _lib.SC_PLAN_IO_BF16), ("mx3", _lib.SC_PLAN_IO_BF16 | _lib.SC_PLAN_MX_FFT_3TERM),
                  ("valu", _lib.SC_PLAN_IO_BF16 | _lib.SC_PLAN_NO_MX_FFT))
        for tag, fl in routes[:3 if H * n_img <= 512 else 2]:
            plan = lib.plan_create([H, 256], [Mx, My], flags=fl)
            assert lib.plan_kernel_name(plan, 0) == ("k_fft2d_fwd_mx" if tag.startswith("mx") else "k_fft2d_fwd3")
            xh = torch.full((n_img, Mx, My, 2), float("nan"))
            lib.transform_forward(plan, mode, x.data_ptr(), xh.data_ptr(), n_img, 0)
            got[tag] = torch.view_as_complex(xh).numpy()
            lib.plan_destroy(plan)
        ref = _ref(xf, H, Mx, My, mode)
        for tag, g in got.items():
            e = np.linalg.norm(g - ref) / np.linalg.norm(ref)
            assert e < (3e-6 if tag == "mx" else 1e-6), (tag, mode, e)


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


def _ref_inv(yh, H, Mx, My, mode, bias):
    # mode SC_INV_PADDED: the kept block zero-padded into the half spectrum, unscaled irfft2 (the C2R ignores the
    # imaginary parts of the DC column), + bias; SC_INV_ADJ_R2C: the adjoint of rfft2(norm="forward"): interior columns
    # count half, 1 / (H W)
    n = yh.shape[0]
    full = np.zeros((n, H, 129), dtype=np.complex128)
    rows = np.r_[np.arange(H - Mx // 2, H), np.arange(0, Mx - Mx // 2)] if Mx > 1 else np.array([0])
    full[:, rows, :My] = yh
    if mode == _lib.SC_INV_ADJ_R2C:
        s = np.full(129, 0.5)
        s[0] = 1.0
        full = full * s / (H * 256)
    y = np.fft.irfft2(full, s=(H, 256), axes=(-2, -1)) * (H * 256)
    return y + (0.0 if bias is None else bias[:, None, None])


@pytest.mark.parametrize("H,Mx,My,n_img", [(256, 64, 33, 3), (256, 20, 9, 3), (128, 64, 33, 5), (128, 12, 33, 4),
                                            (64, 64, 33, 3)])
def test_mx_inverse_vs_float64_and_valu(lib, H, Mx, My, n_img):
    """k_fft2d_inv_mx (round 5, session 2: the row pass of the inverse-type transform writing bfloat16 on the matrix
    cores) against a float64 transform -- the bf16 rounding of the store is the only visible error -- and against
    k_fft2d_inv3<H, sc_bf16>: the two may differ where the fp32 value sits on a rounding boundary (one bf16 ulp)."""
    torch.manual_seed(H + Mx + n_img)
    yh = torch.randn(n_img, Mx, My, 2)
    bias = torch.randn(n_img)                              # channels = n_img: one bias value per image
    yc = torch.view_as_complex(yh).numpy().astype(np.complex128)
    for mode, b in ((_lib.SC_INV_PADDED, bias), (_lib.SC_INV_ADJ_R2C, None)):
        got = {}
        routes = (("mx", _lib.SC_PLAN_IO_BF16), ("valu", _lib.SC_PLAN_IO_BF16 | _lib.SC_PLAN_NO_MX_FFT))
        for tag, fl in routes[:2 if H * n_img <= 512 else 1]:
            plan = lib.plan_create([H, 256], [Mx, My], flags=fl)
            assert lib.plan_kernel_name(plan, 1) == ("k_fft2d_inv_mx" if tag == "mx" else "k_fft2d_inv3")
            y = torch.full((n_img, H, 256), float("nan")).bfloat16()
            lib.transform_inverse(plan, mode, yh.data_ptr(), 0 if b is None else b.data_ptr(), n_img, y.data_ptr(),
                                  n_img, 0)
            got[tag] = y.float().numpy().astype(np.float64)
            lib.plan_destroy(plan)
        ref = _ref_inv(yc, H, Mx, My, mode, None if b is None else b.numpy().astype(np.float64))
        ref_b = torch.from_numpy(ref).float().bfloat16().float().numpy().astype(np.float64)
        for tag, g in got.items():
            assert np.isfinite(g).all(), (tag, mode)
            e = np.linalg.norm(g - ref) / np.linalg.norm(ref)
            assert e < 3e-3, (tag, mode, e)                # bf16 storage: 2^-9 relative per value
            # ... and nothing but that rounding: against the float64 result rounded the same way, at most one ulp (values near zero:
            # the arithmetic's error, relative to the row's magnitude), rarely
            d = np.abs(g - ref_b)
            assert (d <= np.abs(ref_b) * 2.0 ** -7 + 1e-4 * np.abs(ref).max()).all(), (tag, mode, d.max())
            assert (d > 0).mean() < 0.02, (tag, mode, (d > 0).mean())
        if "valu" in got:
            assert (got["mx"] != got["valu"]).mean() < 0.02


def test_mx_inverse_sharded_layout(lib):
    H, Mx, My, n_img, P = 128, 64, 33, 3, 8
    rows = Mx // P
    torch.manual_seed(10)
    yh = torch.randn(n_img, Mx, My, 2)
    plan = lib.plan_create([H, 256], [Mx, My], flags=_lib.SC_PLAN_IO_BF16)
    assert lib.plan_kernel_name(plan, 1) == "k_fft2d_inv_mx"
    y0 = torch.zeros(n_img, H, 256).bfloat16()
    lib.transform_inverse(plan, _lib.SC_INV_PADDED, yh.data_ptr(), 0, 1, y0.data_ptr(), n_img, 0)
    buf = yh.unflatten(1, (P, rows)).movedim(1, 0).contiguous()
    sh = lib.shards(P, rows, n_img * rows * My)
    y1 = torch.full((n_img, H, 256), float("nan")).bfloat16()
    lib.transform_inverse_sharded(plan, _lib.SC_INV_PADDED, buf.data_ptr(), 0, 1, y1.data_ptr(), n_img, sh, 0)
    assert torch.equal(y0.view(torch.int16), y1.view(torch.int16))
    lib.plan_destroy(plan)
