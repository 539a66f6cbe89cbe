"""CPU tier: the factorised last-two-axes kernels for 128 x 128 planes (sc_kernels_plane.h) in host emulation, all
four transform modes through the C-ABI, 2-D and 3-D plans, against numpy's FFT of the same definition (rfftn
restricted to the centred kept block / irfftn of the zero-padded block, spectral_convolution.py:443-449, 500-519,
531-568, and their adjoints)."""
import numpy as np
import pytest
import torch

from neuraloperator_amd import _lib
from engine_runner import emu_lib, rel_l2

TOL = 2e-6


@pytest.fixture(scope="module")
def lib():
    return emu_lib()


def _weights(J, weighted):
    w = np.full(J, 2.0 if weighted else 1.0)
    w[0] = 1.0
    return w


def _ref_forward(x, kept, scale, weighted):
    nd = len(kept)
    J = kept[-1]
    full = np.fft.rfft(x.astype(np.float64), axis=-1)[..., :J]
    for d in range(nd - 1):
        ax = x.ndim - nd + d
        full = np.fft.fft(full, axis=ax)
        rows = [(r - kept[d] // 2) % x.shape[ax] for r in range(kept[d])]
        full = np.take(full, rows, axis=ax)
    return full * scale * (_weights(J, True) if weighted else 1.0)


def _ref_inverse(yhat, spatial, scale, weighted):
    nd = len(spatial)
    kept = yhat.shape[-nd:]
    J = kept[-1]
    cur = yhat.astype(np.complex128) * _weights(J, weighted)
    for d in range(nd - 1):
        ax = yhat.ndim - nd + d
        shp = list(cur.shape)
        shp[ax] = spatial[d]
        spec = np.zeros(shp, dtype=np.complex128)
        idx = [slice(None)] * cur.ndim
        for r in range(kept[d]):
            idx[ax] = (r - kept[d] // 2) % spatial[d]
            src = [slice(None)] * cur.ndim
            src[ax] = r
            spec[tuple(idx)] = cur[tuple(src)]
        cur = np.fft.ifft(spec, axis=ax) * spatial[d]
    n = np.arange(spatial[-1])
    ph = np.exp(2j * np.pi * np.outer(np.arange(J), n) / spatial[-1])
    return np.real(cur @ ph) * scale


CASES = [
    ((128, 128), (32, 17), 3),          # 2-D grid: one plane per image
    ((128, 128), (21, 9), 2),           # odd kept rows, fewer columns
    ((6, 128, 128), (4, 32, 17), 2),    # 3-D: planes + axis pass over the first dim (FNO3d 128^3 shape family)
    ((128, 128), (1, 1), 1),
    ((128, 128, 128), (21, 32, 17), 1),  # first axis on the 128-point line kernel as well (k_ax128)
]


@pytest.mark.parametrize("spatial,kept,n_img", CASES, ids=["x".join(map(str, s)) + "_k" + "x".join(map(str, k)) for s, k, _ in CASES])
@pytest.mark.parametrize("norm", ["forward", "ortho"])
def test_plane_kernels(lib, spatial, kept, n_img, norm):
    if norm == "ortho" and len(spatial) == 3:
        pytest.skip("one norm is enough for the 3-D case")
    rng = np.random.default_rng(7)
    plan = lib.plan_create(list(spatial), list(kept), fft_norm=norm, flags=0)
    ntot = int(np.prod(spatial))
    sf, si = (1.0 / ntot, 1.0) if norm == "forward" else (ntot ** -0.5, ntot ** -0.5)
    try:
        assert lib.plan_kernel_name(plan, 0) == "k_pl128_fwd" and lib.plan_kernel_name(plan, 1) == "k_pl128_inv"
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


def test_plane_scope(lib):
    """Other plane sizes / larger kept blocks stay on the size-agnostic kernels."""
    for spatial, kept in (((128, 128), (64, 17)), ((128, 128), (32, 33)), ((64, 128), (32, 17)), ((128, 64), (32, 17))):
        plan = lib.plan_create(list(spatial), list(kept), fft_norm="forward", flags=0)
        try:
            assert lib.plan_kernel_name(plan, 0) != "k_pl128_fwd"
        finally:
            lib.plan_destroy(plan)
