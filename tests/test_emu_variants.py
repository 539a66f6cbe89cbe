"""CPU tier: the variants that need frequency maps or complex passes -- resolution-changing inverse,
complex data, the spectral skip-path resample -- through the C-ABI stage entry points of the
host-emulation build, against golden vectors of the verbatim reference (oracle/gen_golden.py).
The maps come from neuraloperator_amd/modes.py, i.e. this also pins that host logic."""
import json

import numpy as np
import pytest
import torch

from conftest import golden_names, load_golden
from engine_runner import emu_lib, rel_l2, staged_fwd_bwd
from neuraloperator_amd import _lib, modes

TOL = 1e-5
RES = [n for n in golden_names() if n.startswith("res_")]
CPLX = [n for n in golden_names() if n.startswith("cplx_")]


@pytest.fixture(scope="module")
def lib():
    return emu_lib()


def _block(g, kept, w_start):
    w = torch.from_numpy(g["param_0"])                       # dense stored weight
    idx = (slice(None), slice(None)) + tuple(slice(s, s + k) for s, k in zip(w_start, kept))
    return w, idx


@pytest.mark.parametrize("passes", ["mdft", "valu"])
@pytest.mark.parametrize("name", RES + CPLX)
def test_staged_variants_match_golden(lib, name, passes):
    g = load_golden(name)
    kw = json.loads(str(g["ctor_kwargs"]))
    cplx = bool(kw.get("complex_data", False))
    x, gy = torch.from_numpy(g["x"]), torch.from_numpy(g["g"])
    bias = torch.from_numpy(g["bias"])
    spatial, out_spatial = list(x.shape[2:]), list(g["y"].shape[2:])
    nm, mx = list(g["n_modes_attr"]), list(g["max_n_modes_attr"])
    kept, w_start = (modes.kept_block_complex if cplx else modes.kept_block)(spatial, nm, mx)
    w, idx = _block(g, kept, w_start)
    fa = modes.analysis_freqs(spatial, kept, cplx)
    fs, real_col = modes.synthesis_freqs(spatial, out_spatial, kept, cplx)
    flags = _lib.SC_PLAN_NO_MDFT if passes == "valu" else 0
    y, gx, gwb, gb = staged_fwd_bwd(lib, x, w[idx], bias, gy, kept, out_spatial, fa, fs, real_col,
                                    complex_data=cplx, flags=flags)
    gw = torch.zeros_like(w)
    gw[idx] = gwb
    assert rel_l2(y.numpy(), g["y"]) < TOL
    assert rel_l2(gx.numpy(), g["gx"]) < TOL
    assert rel_l2(gw.numpy(), g["g_param_0"]) < TOL
    assert rel_l2(gb.numpy(), g["gbias"]) < TOL


@pytest.mark.parametrize("name", [n for n in golden_names() if n.startswith("xform_3d")])
def test_spectral_resample_matches_golden(lib, name):
    """SpectralConv.transform for 3-d and up = resample.py:54-66: truncated forward transform on the old
    grid, zero-padded inverse on the new one, both with the resample's own row convention."""
    g = load_golden(name)
    x = torch.from_numpy(g["x"])
    spatial, out_spatial = list(x.shape[2:]), [int(v) for v in g["output_shape"]]
    kept, fa, fs = modes.resample_block(spatial, out_spatial)
    b, c = x.shape[:2]
    pa = lib.plan_create(spatial, kept, freq=fa)
    pb = lib.plan_create(out_spatial, kept, freq=fs)
    xhat = torch.empty(b, c, *kept, 2)
    ws = torch.empty(max(lib.plan_workspace_bytes(pa, b * c), lib.plan_workspace_bytes(pb, b * c), 256), dtype=torch.uint8)
    lib.transform_forward(pa, _lib.SC_FWD_SCALED, x.contiguous().data_ptr(), xhat.data_ptr(), b * c, ws.data_ptr(), 0)
    t = torch.empty(b, c, *out_spatial)
    lib.transform_inverse(pb, _lib.SC_INV_PADDED, xhat.data_ptr(), 0, c, t.data_ptr(), b * c, ws.data_ptr(), 0)
    lib.plan_destroy(pa)
    lib.plan_destroy(pb)
    assert rel_l2(t.numpy(), g["t"]) < TOL


def test_frequency_map_validation(lib):
    with pytest.raises(_lib.EngineError):
        lib.plan_create([8, 8], [4, 3], freq=[[0, 1, 2, 9], None])          # index outside the grid
    with pytest.raises(_lib.EngineError):
        lib.plan_create([8, 8], [4, 3], freq=[None, [0, 1, 6]])             # real data: last dim beyond n/2
    with pytest.raises(_lib.EngineError):
        lib.plan_create([8, 8], [4, 3], freq=[[0, 1], None])                # wrong length
    p = lib.plan_create([8, 8], [4, 3], freq=[[6, 7, 0, None], None])       # a dropped row is fine
    lib.plan_destroy(p)


@pytest.mark.parametrize("passes", ["mdft", "valu"])
@pytest.mark.parametrize("n,k,lines", [(16, 8, 5), (12, 5, 3), (64, 32, 9), (10, 10, 2)])
def test_one_complex_axis_with_centred_rows(lib, n, k, lines, passes):
    """The sharded-dim pass of mpu.SpatialParallelSpectralConv (EngineOps.forward_axis / inverse_axis): a 1-d
    complex plan whose kept row r reads / lands at FFT index (r - k//2) mod n, forward scaled by 1/n
    (fft_norm="forward"), inverse zero-padded and unscaled -- and the SC_FWD_ADJ / SC_INV_ADJ pair autograd uses."""
    from neuraloperator_amd.mpu.spatial_parallel import centred_rows
    torch.manual_seed(n + k)
    rows = centred_rows(k, n)
    flags = _lib.SC_PLAN_COMPLEX | (_lib.SC_PLAN_NO_MDFT if passes == "valu" else 0)
    plan = lib.plan_create([n], [k], flags=flags, freq=[rows])
    try:
        b = 2
        x = torch.randn(b, lines, n, dtype=torch.cfloat)
        xh = torch.empty(b, lines, k, dtype=torch.cfloat)
        ws = torch.empty(max(lib.plan_workspace_bytes(plan, b * lines), 256), dtype=torch.uint8)
        ix = torch.as_tensor(rows)
        lib.transform_forward(plan, _lib.SC_FWD_SCALED, torch.view_as_real(x).data_ptr(),
                              torch.view_as_real(xh).data_ptr(), b * lines, ws.data_ptr(), 0)
        ref = torch.fft.fft(x, dim=-1, norm="forward").index_select(-1, ix)
        assert rel_l2(xh.numpy(), ref.numpy()) < TOL
        yh = torch.randn(b, lines, k, dtype=torch.cfloat)
        y = torch.empty(b, lines, n, dtype=torch.cfloat)
        lib.transform_inverse(plan, _lib.SC_INV_PADDED, torch.view_as_real(yh).data_ptr(), 0, lines,
                              torch.view_as_real(y).data_ptr(), b * lines, ws.data_ptr(), 0)
        z = torch.zeros(b, lines, n, dtype=torch.cfloat).index_add(-1, ix, yh)
        assert rel_l2(y.numpy(), torch.fft.ifft(z, dim=-1, norm="forward").numpy()) < TOL
        # adjoints (what TransformForwardFn / TransformInverseFn launch in backward): <A x, v> = <x, A^H v>
        gx = torch.empty_like(x)
        lib.transform_inverse(plan, _lib.SC_INV_ADJ_R2C, torch.view_as_real(yh).data_ptr(), 0, lines,
                              torch.view_as_real(gx).data_ptr(), b * lines, ws.data_ptr(), 0)
        lhs = torch.sum(torch.conj(yh) * ref)
        rhs = torch.sum(torch.conj(gx) * x)
        assert abs(lhs - rhs) < 1e-4 * max(abs(lhs), 1.0)
        gh = torch.empty_like(yh)
        lib.transform_forward(plan, _lib.SC_FWD_ADJ_C2R, torch.view_as_real(x).data_ptr(),
                              torch.view_as_real(gh).data_ptr(), b * lines, ws.data_ptr(), 0)
        lhs = torch.sum(torch.conj(x) * torch.fft.ifft(z, dim=-1, norm="forward"))
        rhs = torch.sum(torch.conj(gh) * yh)
        assert abs(lhs - rhs) < 1e-4 * max(abs(lhs), 1.0)
    finally:
        lib.plan_destroy(plan)
