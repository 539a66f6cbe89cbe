"""CPU tier: the engine's kernels, compiled for the host by tests/emu (same source as the
HIP build), against the golden vectors of the verbatim reference.  This validates index
maths / LDS choreography / barrier placement before any GPU time is spent; the parity
tests proper are the ``-m gpu`` ones."""
import numpy as np
import pytest
import torch

from conftest import DENSE_GOLDEN, load_golden
from engine_runner import emu_lib, layer_fwd_bwd, rel_l2
from neuraloperator_amd import _lib

TOL = 1e-5  # north-star parity bar (fp32 rel-L2)


@pytest.fixture(scope="module")
def lib():
    return emu_lib()


SMALL = [n for n in DENSE_GOLDEN if n not in ("darcy_c1_16x16_m12_c32", "d2_64x64_m16_c8", "d3_16x16x16_m8")]


@pytest.mark.parametrize("passes", ["mdft", "valu"])
@pytest.mark.parametrize("name", SMALL)
def test_generic_path_matches_golden(lib, name, passes):
    """size-agnostic passes: on the matrix cores (k_mdft_*, emulated MFMA) and on the VALU kernels"""
    g = load_golden(name)
    x, w, b, gy = (torch.from_numpy(g[k]) for k in ("x", "weight", "bias", "g"))
    flags = _lib.SC_PLAN_FORCE_GENERIC | (_lib.SC_PLAN_NO_MDFT if passes == "valu" else 0)
    y, gx, gw, gb, _ = layer_fwd_bwd(lib, x, w, b, gy, list(g["n_modes_attr"]), list(g["max_n_modes_attr"]),
                                     flags=flags)
    assert rel_l2(y.numpy(), g["y"]) < TOL
    assert rel_l2(gx.numpy(), g["gx"]) < TOL
    assert rel_l2(gw.numpy(), g["gw"]) < TOL
    assert rel_l2(gb.numpy(), g["gbias"]) < TOL


# ------------------------------------------------------------------------------------------
# fused power-of-two FFT kernels (sc_kernels_fft.h) in emulation, against the CPU oracle
# ------------------------------------------------------------------------------------------
FAST_CASES = [
    # B, Cin, Cout, H, n_modes
    (1, 2, 2, 256, (64, 64)),     # the BASELINE metric geometry (one image per workgroup)
    (1, 1, 2, 64, (64, 64)),      # single row group, all 64 row frequencies kept
    (1, 2, 1, 128, (32, 32)),
    (1, 1, 1, 512, (10, 6)),
    (2, 1, 1, 256, (5, 7)),       # odd kept counts
    # more images than the emulated chip holds workgroups (sc_cu_count() = 1 in emulation): the persistent kernels
    # walk several images per workgroup with their prefetch running across them
    (4, 2, 2, 64, (20, 16)),
    (2, 2, 3, 128, (12, 12)),
]


@pytest.mark.parametrize("gen", [3, 2], ids=["gen3", "gen2"])
@pytest.mark.parametrize("case", FAST_CASES, ids=lambda c: f"H{c[3]}_m{c[4][0]}x{c[4][1]}")
def test_fused_fft_path_matches_oracle(lib, case, gen):
    from oracle import spectral_oracle as so
    from neuraloperator_amd.modes import halve_last_mode, kept_block

    b, ci, co, H, modes = case
    torch.manual_seed(11)
    nm = halve_last_mode(modes)
    kept, _ = kept_block([H, 256], nm, nm)
    flags = _lib.SC_PLAN_FFT_GEN2 if gen == 2 else 0
    plan = lib.plan_create([H, 256], kept, flags=flags)
    assert lib.plan_is_fast(plan)
    assert lib.plan_kernel_name(plan, 0) == ("k_fft2d_fwd" if gen == 2 else "k_fft2d_fwd3")
    lib.plan_destroy(plan)
    x = torch.randn(b, ci, H, 256)
    w = torch.empty(ci, co, *nm, dtype=torch.cfloat).normal_(0, 0.5)
    bias = torch.randn(co, 1, 1)
    g = torch.randn(b, co, H, 256)
    xc, wc, bc = x.clone().requires_grad_(True), w.clone().requires_grad_(True), bias.clone().requires_grad_(True)
    yo = so.forward_torch(xc, wc, bc, nm, nm)
    yo.backward(g)
    y, gx, gw, gb, xh = layer_fwd_bwd(lib, x, w, bias, g, nm, nm, flags=flags)
    assert rel_l2(y.numpy(), yo.detach().numpy()) < TOL
    assert rel_l2(gx.numpy(), xc.grad.numpy()) < TOL
    assert rel_l2(gw.numpy(), wc.grad.numpy()) < TOL
    assert rel_l2(gb.numpy(), bc.grad.numpy()) < TOL
    # the saved truncated spectrum itself (weight order) against the fp64 kept-rows oracle
    _, xk = so.forward_np64(x.numpy(), w.numpy(), bias.numpy(), nm, nm)
    assert rel_l2(xh.numpy(), xk) < TOL


def bf16_checks(y, y_ref32, what):
    """y: bfloat16 result; y_ref32: the fp32 oracle result on the same (bf16-valued) inputs.  The engine rounds
    its fp32 result once (nearest even), so it sits within half a bf16 ulp (2^-9 relative) of the fp32 oracle
    plus fp32 round-off, and agrees bit for bit with the rounded oracle except where the two fp32 values
    straddle a rounding boundary."""
    assert y.dtype == torch.bfloat16, what
    yf = y.float()
    err = (yf - y_ref32).abs()
    bound = y_ref32.abs() * 2.0 ** -8 + 1e-5 * y_ref32.abs().max()
    assert bool((err <= bound).all()), f"{what}: {float((err - bound).max())} past one bf16 ulp"
    same = (y.view(torch.int16) == y_ref32.bfloat16().view(torch.int16)).float().mean().item()
    assert same > 0.98, f"{what}: only {same:.4f} of the values equal the rounded oracle bit for bit"


@pytest.mark.parametrize("case", [(1, 2, 2, 256, (64, 64)), (1, 1, 2, 64, (20, 16)), (2, 1, 1, 128, (5, 7))],
                         ids=lambda c: f"H{c[3]}_m{c[4][0]}x{c[4][1]}")
def test_fused_fft_bf16_io_matches_oracle(lib, case):
    """SC_PLAN_IO_BF16: x / gy read as bfloat16, y / gx stored as bfloat16 (nearest even), arithmetic and
    the fp32 outputs (saved spectrum, gW, gbias) as in the fp32 path on the same input values."""
    from oracle import spectral_oracle as so
    from neuraloperator_amd.modes import halve_last_mode, kept_block

    b, ci, co, H, modes = case
    torch.manual_seed(12)
    nm = halve_last_mode(modes)
    kept, _ = kept_block([H, 256], nm, nm)
    with pytest.raises(_lib.EngineError):             # size-agnostic passes have no bf16 I/O: the host converts
        lib.plan_create([48, 40], [8, 5], flags=_lib.SC_PLAN_IO_BF16)
    with pytest.raises(_lib.EngineError):
        lib.plan_create([H, 256], kept, flags=_lib.SC_PLAN_IO_BF16 | _lib.SC_PLAN_FFT_GEN2)
    # round 5: forward-type transforms of bf16 tensors run their row pass on the matrix cores (k_fft2d_fwd_mx: the bf16
    # input is exact in the MFMA's input format, the twiddles are three bf16 terms); SC_PLAN_NO_MX_FFT = the
    # vector-ALU kernel.  Both against the same oracle, the matrix-core one no further from the fp64 spectrum
    for fl, name in ((_lib.SC_PLAN_IO_BF16, "k_fft2d_fwd_mx"), (_lib.SC_PLAN_IO_BF16 | _lib.SC_PLAN_NO_MX_FFT, "k_fft2d_fwd3")):
        plan = lib.plan_create([H, 256], kept, flags=fl)
        assert lib.plan_kernel_name(plan, 0) == name
        lib.plan_destroy(plan)
    x = torch.randn(b, ci, H, 256).bfloat16()
    w = torch.empty(ci, co, *nm, dtype=torch.cfloat).normal_(0, 0.5)
    bias = torch.randn(co, 1, 1)
    g = torch.randn(b, co, H, 256).bfloat16()
    xc, wc, bc = x.float().requires_grad_(True), w.clone().requires_grad_(True), bias.clone().requires_grad_(True)
    yo = so.forward_torch(xc, wc, bc, nm, nm)
    yo.backward(g.float())
    _, xk = so.forward_np64(x.float().numpy(), w.numpy(), bias.numpy(), nm, nm)
    errs = {}
    for fl in (_lib.SC_PLAN_IO_BF16, _lib.SC_PLAN_IO_BF16 | _lib.SC_PLAN_NO_MX_FFT)[:2 if H <= 64 else 1]:   # (CPU-tier time)
        y, gx, gw, gb, xh = layer_fwd_bwd(lib, x, w, bias, g, nm, nm, flags=fl)
        bf16_checks(y, yo.detach(), "y")
        bf16_checks(gx, xc.grad, "gx")
        assert rel_l2(gw.numpy(), wc.grad.numpy()) < TOL
        assert rel_l2(gb.numpy(), bc.grad.numpy()) < TOL
        errs[fl] = rel_l2(xh.numpy(), xk)
        assert errs[fl] < TOL
    assert errs[_lib.SC_PLAN_IO_BF16] < 2e-6, errs           # fp32 round-off class (the vector-ALU kernel: ~3e-7)


def test_emu_library_exports_and_errors(lib):
    for s in _lib.ScEngineLib.SYMBOLS:
        assert hasattr(lib.lib, s)
    with pytest.raises(_lib.EngineError):
        lib.plan_create([16, 16], [17, 9])          # more modes than the spectrum has
    with pytest.raises(_lib.EngineError):
        lib.plan_create([16] * 5, [4] * 5)


@pytest.mark.parametrize("width", [64, 48, 40])
def test_mdft_tail_column_matches_oracle(lib, width):
    """last axis keeps 2^k + 1 columns (J = 17): the matrix-core pass does 16 columns as one MFMA tile
    and the 17th as a VALU dot product.  Width 64 takes the LDS-staged kernel (k_mdft_r2c_lds<.., TAIL>),
    48 and 40 (not multiples of 32) the LDS-staged one with its table in global memory (k_mdft_r2c_stage<.., TAIL>)."""
    from oracle import spectral_oracle as so
    from neuraloperator_amd.modes import halve_last_mode
    torch.manual_seed(5)
    spatial, modes = (12, width), (6, 32)
    nm = halve_last_mode(modes)
    assert nm[-1] == 17
    x = torch.randn(2, 3, *spatial)
    w = torch.empty(3, 2, *nm, dtype=torch.cfloat).normal_(0, 0.5)
    bias = torch.randn(2, 1, 1)
    g = torch.randn(2, 2, *spatial)
    xc, wc, bc = x.clone().requires_grad_(True), w.clone().requires_grad_(True), bias.clone().requires_grad_(True)
    yo = so.forward_torch(xc, wc, bc, nm, nm)
    yo.backward(g)
    plan = lib.plan_create(list(spatial), list(nm))
    assert lib.plan_kernel_name(plan, 0) == ("k_mdft_r2c_lds" if width % 32 == 0 else "k_mdft_r2c_stage")
    assert lib.plan_kernel_name(plan, 1) == "k_mdft_c2r_lds"
    lib.plan_destroy(plan)
    y, gx, gw, gb, _ = layer_fwd_bwd(lib, x, w, bias, g, nm, nm, flags=0)
    assert rel_l2(y.numpy(), yo.detach().numpy()) < TOL
    assert rel_l2(gx.numpy(), xc.grad.numpy()) < TOL
    assert rel_l2(gw.numpy(), wc.grad.numpy()) < TOL
    assert rel_l2(gb.numpy(), bc.grad.numpy()) < TOL


@pytest.mark.parametrize("spatial,modes,b,kern", [
    ((21, 85), (8, 32), 2, "k_mdft_r2c_stage"), ((13, 141), (6, 64), 1, "k_mdft_r2c_stage"),
    ((45, 53), (12, 16), 3, "k_mdft_r2c_stage"), ((37, 421), (10, 32), 1, "k_mdft_r2c_stage"),
    ((5, 7, 43), (4, 4, 10), 2, "k_mdft_r2c_stage"), ((150, 85), (8, 32), 1, "k_mdft_r2c_stage"),
    ((13, 141), (6, 80), 2, "k_mdft_r2c"), ((9, 211), (4, 140), 1, "k_mdft_r2c"),
    # few columns under a long first axis: the four waves of a block split its rows (k_mdft_axis<.., KS>)
    ((70, 9), (8, 4), 2, "k_mdft_r2c_stage"), ((131, 13), (40, 6), 1, "k_mdft_r2c_stage")],
    ids=["21x85_tail", "13x141_m64_tail", "45x53", "37x421_tail", "5x7x43", "150x85_two_tiles", "13x141_m80", "9x211_m140",
         "70x9_ksplit", "131x13_ksplit_4jt"])
@pytest.mark.parametrize("span", [True, False], ids=["span", "chunked"])
def test_mdft_ragged_width_matches_oracle(lib, spatial, modes, b, kern, span):
    """widths that are not a multiple of 8 (the reference's Darcy grids: 85 / 141 / 211 / 421) on the matrix cores:
    k_mdft_r2c_stage brings 128-line tiles through LDS with row-wise 4-byte loads (<= 2 column tiles of kept modes),
    k_mdft_r2c<.., RAGGED> (more kept modes) pads the last group of 8 points with zeros; k_mdft_c2r_stage parks a
    128-line tile of the spectrum in LDS and streams the table (<= 36 kept columns), k_mdft_c2r (more) looks the bias
    up per row when an image's line count (here odd) is not a multiple of a wave's 32 RT lines."""
    from oracle import spectral_oracle as so
    from neuraloperator_amd.modes import halve_last_mode
    torch.manual_seed(7)
    nm = halve_last_mode(modes)
    nd = len(spatial)
    x = torch.randn(b, 3, *spatial)
    w = torch.empty(3, 2, *nm, dtype=torch.cfloat).normal_(0, 0.5)
    bias = torch.randn(2, *([1] * nd))
    g = torch.randn(b, 2, *spatial)
    xc, wc, bc = x.clone().requires_grad_(True), w.clone().requires_grad_(True), bias.clone().requires_grad_(True)
    yo = so.forward_torch(xc, wc, bc, nm, nm)
    yo.backward(g)
    # span: the inverse pass in 32-line blocks whose N-line span is staged whole (k_mdft_c2r_span); chunked: SC_PLAN_NO_SPAN
    flags = 0 if span else _lib.SC_PLAN_NO_SPAN
    plan = lib.plan_create(list(spatial), list(nm), flags=flags)
    assert lib.plan_kernel_name(plan, 0) == kern
    k1 = kern.replace("r2c", "c2r")
    assert lib.plan_kernel_name(plan, 1) == (k1.replace("_stage", "_span") if span else k1)
    lib.plan_destroy(plan)
    y, gx, gw, gb, _ = layer_fwd_bwd(lib, x, w, bias, g, nm, nm, flags=flags)
    assert rel_l2(y.numpy(), yo.detach().numpy()) < TOL
    assert rel_l2(gx.numpy(), xc.grad.numpy()) < TOL
    assert rel_l2(gw.numpy(), wc.grad.numpy()) < TOL
    assert rel_l2(gb.numpy(), bc.grad.numpy()) < TOL
    # an odd storage offset (4-byte aligned view): same kernels, same numbers
    buf = torch.zeros(x.numel() + 1)
    xo = buf[1:].view_as(x).copy_(x)
    y2, _, _, _, _ = layer_fwd_bwd(lib, xo, w, bias, g, nm, nm, flags=flags)
    assert rel_l2(y2.numpy(), yo.detach().numpy()) < TOL


@pytest.mark.parametrize("case", [((128, 128), (32, 32), 2, 2, 3), ((128, 64), (40, 16), 1, 3, 2),
                                  ((3, 128, 32), (2, 12, 8), 2, 2, 2), ((128, 128), (128, 64), 1, 1, 2),
                                  ((64, 64), (24, 32), 1, 3, 2), ((64, 32), (12, 8), 3, 1, 1),
                                  ((3, 32, 64), (3, 16, 16), 1, 2, 2), ((5, 64, 32), (4, 30, 16), 1, 1, 1)],
                         ids=["128x128_m32", "128x64_m40x16", "3x128x32", "128x128_allrows",
                              "64x64_m24", "64x32_m12_3planes", "3x32x64", "5x64x32_odd_planes"])
def test_plane_kernels_match_oracle(lib, case):
    """second-to-last axis of 128 rows: the last two axes run in ONE launch each way (k_mdft_r2c_lds<.., JP>,
    k_mdft_c2r_lds<.., PLANE>) -- 1, 2 and 4 row tiles, with and without the VALU tail column, 2-D and 3-D."""
    from oracle import spectral_oracle as so
    from neuraloperator_amd.modes import halve_last_mode
    spatial, modes, b, ci, co = case
    torch.manual_seed(9)
    nm = halve_last_mode(modes)
    x = torch.randn(b, ci, *spatial)
    w = torch.empty(ci, co, *nm, dtype=torch.cfloat).normal_(0, 0.5)
    bias = torch.randn(co, *(1,) * len(spatial))
    g = torch.randn(b, co, *spatial)
    xc, wc, bc = x.clone().requires_grad_(True), w.clone().requires_grad_(True), bias.clone().requires_grad_(True)
    yo = so.forward_torch(xc, wc, bc, nm, nm)
    yo.backward(g)
    # 128 x 128 planes with a kept block <= 32 x 17 have their own factorised kernels (sc_kernels_plane.h, tested in
    # test_emu_plane128.py); SC_PLAN_FORCE_GENERIC keeps such a plan on the direct-DFT plane kernels tested here
    # (session 2: the same for 64 x 64 planes, sc_kernels_plane64.h / test_emu_plane64.py)
    fft_plane = list(spatial[-2:]) in ([128, 128], [64, 64]) and nm[-2] <= 32 and nm[-1] <= 17
    flags = _lib.SC_PLAN_FORCE_GENERIC if fft_plane else 0
    plan = lib.plan_create(list(spatial), list(nm), flags=flags)
    fused = 2 * nm[-2] <= spatial[-2]
    assert (lib.plan_kernel_name(plan, 0) == "k_mdft_r2c_lds<plane>") == fused      # > 64 kept rows: separate passes
    assert (lib.plan_kernel_name(plan, 1) == "k_mdft_c2r_lds<plane>") == fused
    lib.plan_destroy(plan)
    y, gx, gw, gb, _ = layer_fwd_bwd(lib, x, w, bias, g, nm, nm, flags=flags)
    assert rel_l2(y.numpy(), yo.detach().numpy()) < TOL
    assert rel_l2(gx.numpy(), xc.grad.numpy()) < TOL
    assert rel_l2(gw.numpy(), wc.grad.numpy()) < TOL
    assert rel_l2(gb.numpy(), bc.grad.numpy()) < TOL
