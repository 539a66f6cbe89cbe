"""CPU tier: the two-pass factorised transforms for large power-of-two grids (sc_kernels_fft2p.h) in host emulation:
all four transform modes through the C-ABI against numpy's FFT of the same definition (rfft2 restricted to the kept
block / irfft2 of the zero-padded block, spectral_convolution.py:443-449, 500-519, 531-568, and their adjoints) and
against the size-agnostic direct-DFT route of the same library (SC_PLAN_FORCE_GENERIC)."""
import numpy as np
import pytest
import torch

from neuraloperator_amd import _lib
from engine_runner import emu_lib, rel_l2

TOL = 2e-6


@pytest.fixture(scope="module")
def lib():
    return emu_lib()


def _ref_forward(x, kept, scale, weighted):
    """kept block of the centred spectrum: rows f = r - K0/2, columns 0..J-1; `weighted` = adjoint of irfft2."""
    K0, J = kept
    N0, N1 = x.shape[-2:]
    full = np.fft.fft(np.fft.rfft(x.astype(np.float64), axis=-1)[..., :J], axis=-2)
    rows = [(r - K0 // 2) % N0 for r in range(K0)]
    out = full[..., rows, :] * scale
    if weighted:
        w = np.full(J, 2.0)
        w[0] = 1.0
        out = out * w
    return out


def _ref_inverse(yhat, spatial, scale, weighted, bias=None):
    K0, J = yhat.shape[-2:]
    N0, N1 = spatial
    n_img = yhat.shape[0]
    spec = np.zeros((n_img, N0, J), dtype=np.complex128)
    for r in range(K0):
        spec[:, (r - K0 // 2) % N0, :] = yhat[:, r, :]
    cols = np.fft.ifft(spec, axis=-2) * N0                      # unnormalised inverse over rows
    w = np.full(J, 2.0 if weighted else 1.0)
    w[0] = 1.0
    n = np.arange(N1)
    # y[n] = Re sum_j w_j cols[j] e^{+2 pi i j n / N1}: irfft semantics (weighted) or the adjoint of the pruned rfft
    ph = np.exp(2j * np.pi * np.outer(np.arange(J), n) / N1)
    y = np.real(np.einsum("brj,jn->brn", cols * w, ph)) * scale
    if bias is not None:
        y = y + bias[:, None, None]
    return y


CASES = [
    # spatial, kept, images
    ((1024, 1024), (256, 129), 1),      # BASELINE configs[4]
    ((1024, 512), (64, 33), 2),
    ((512, 1024), (31, 17), 2),         # odd kept rows, ragged columns
    ((512, 512), (256, 120), 1),        # half of the rows, widest column range the codelets cover
    ((1024, 1024), (2, 1), 1),
    ((1024, 1024), (101, 129), 2),      # round 4: k_f2p_c2r_w1024 (129 kept columns) behind fewer, odd kept rows
    ((1024, 1024), (256, 129), 5),      # ... and several images (the persistent item loop of the emulated 4-workgroup grid)
    # round 3: lines of 32 P points, P in {2, 3, 4, 5, 6, 8, 10, 12, 20} (radix 3 / 5 codelets, composite P)
    ((64, 64), (32, 17), 3),            # the common small grid (fno2d_64, modes 32): P = 2
    ((96, 96), (24, 13), 2),            # P = 3
    ((192, 192), (64, 33), 1),          # P = 6 = 3 x 2, the widest kept block of the fused 256-wide kernels
    ((160, 320), (20, 11), 2),          # P = 5 and 10 = 5 x 2
    ((384, 640), (48, 25), 1),          # P = 12 = 3 x 4 and 20 = 5 x 4
    ((128, 64), (16, 9), 2),            # P = 4 and 2 (not a 128 x 128 plane)
    ((256, 256), (128, 65), 1),         # P = 8: a kept block beyond the fused kernels' 64 x 33
    ((64, 1024), (5, 3), 2),            # mixed with a 1024-point line
]


@pytest.mark.parametrize("spatial,kept,n_img", CASES, ids=[f"{s[0]}x{s[1]}_k{k[0]}x{k[1]}" for s, k, _ in CASES])
def test_two_pass_transforms(lib, spatial, kept, n_img):
    rng = np.random.default_rng(5)
    N0, N1 = spatial
    K0, J = kept
    # planes below 128 x 128 points stay on the direct-DFT passes by default (faster there): forced here
    small = N0 * N1 < 128 * 128
    if small:
        pd = lib.plan_create(list(spatial), list(kept), fft_norm="forward", flags=0)
        assert lib.plan_kernel_name(pd, 0) != "k_f2p_r2c"
        lib.plan_destroy(pd)
    plan = lib.plan_create(list(spatial), list(kept), fft_norm="forward", flags=_lib.SC_PLAN_F2P_SMALL_ALWAYS if small else 0)
    try:
        assert lib.plan_kernel_name(plan, 0) == "k_f2p_r2c", "grids of 32 P points per axis take the two-pass route"
        ws = torch.empty(max(lib.plan_workspace_bytes(plan, n_img), 256), dtype=torch.uint8)
        x = torch.from_numpy(rng.standard_normal((n_img, N0, N1)).astype(np.float32))
        ntot = N0 * N1
        for mode, scale, weighted in ((_lib.SC_FWD_SCALED, 1.0 / ntot, False), (_lib.SC_FWD_ADJ_C2R, 1.0, True)):
            xhat = torch.full((n_img, K0, J), float("nan"), dtype=torch.complex64)
            lib.transform_forward(plan, mode, x.data_ptr(), torch.view_as_real(xhat).data_ptr(), n_img, ws.data_ptr(), 0)
            ref = _ref_forward(x.numpy(), kept, scale, weighted)
            assert rel_l2(xhat.numpy(), ref) < TOL, f"forward mode {mode}"
        yh = (rng.standard_normal((n_img, K0, J)) + 1j * rng.standard_normal((n_img, K0, J))).astype(np.complex64)
        yhat = torch.from_numpy(yh)
        channels = n_img
        bias = torch.from_numpy(rng.standard_normal(channels).astype(np.float32))
        for mode, scale, weighted, b in ((_lib.SC_INV_PADDED, 1.0, True, bias), (_lib.SC_INV_ADJ_R2C, 1.0 / ntot, False, None)):
            y = torch.full((n_img, N0, N1), float("nan"), dtype=torch.float32)
            lib.transform_inverse(plan, mode, torch.view_as_real(yhat).data_ptr(), 0 if b is None else b.data_ptr(),
                                  channels, y.data_ptr(), n_img, ws.data_ptr(), 0)
            ref = _ref_inverse(yh, spatial, scale, weighted, None if b is None else b.numpy().astype(np.float64))
            assert rel_l2(y.numpy(), ref) < TOL, f"inverse mode {mode}"
    finally:
        lib.plan_destroy(plan)


def test_two_pass_matches_size_agnostic_route(lib):
    """Same plan description on the direct-DFT passes (SC_PLAN_FORCE_GENERIC): the two routes are interchangeable."""
    rng = np.random.default_rng(6)
    spatial, kept, n_img = (512, 512), (24, 33), 2
    x = torch.from_numpy(rng.standard_normal((n_img, *spatial)).astype(np.float32))
    yhat = torch.from_numpy((rng.standard_normal((n_img, *kept)) + 1j * rng.standard_normal((n_img, *kept))).astype(np.complex64))
    bias = torch.from_numpy(rng.standard_normal(2).astype(np.float32))
    res = {}
    for tag, flags in (("f2p", 0), ("generic", _lib.SC_PLAN_FORCE_GENERIC)):
        plan = lib.plan_create(list(spatial), list(kept), fft_norm="ortho", flags=flags)
        try:
            ws = torch.empty(max(lib.plan_workspace_bytes(plan, n_img), 256), dtype=torch.uint8)
            xhat = torch.empty((n_img, *kept), dtype=torch.complex64)
            lib.transform_forward(plan, _lib.SC_FWD_SCALED, x.data_ptr(), torch.view_as_real(xhat).data_ptr(), n_img, ws.data_ptr(), 0)
            y = torch.empty((n_img, *spatial), dtype=torch.float32)
            lib.transform_inverse(plan, _lib.SC_INV_PADDED, torch.view_as_real(yhat).data_ptr(), bias.data_ptr(), 2,
                                  y.data_ptr(), n_img, ws.data_ptr(), 0)
            res[tag] = (xhat.numpy().copy(), y.numpy().copy(), lib.plan_kernel_name(plan, 0))
        finally:
            lib.plan_destroy(plan)
    assert res["f2p"][2] == "k_f2p_r2c" and res["generic"][2] != "k_f2p_r2c"
    assert rel_l2(res["f2p"][0], res["generic"][0]) < TOL
    assert rel_l2(res["f2p"][1], res["generic"][1]) < TOL


def test_two_pass_scope(lib):
    """Outside the route's scope the plan stays on the size-agnostic passes (and still works)."""
    for spatial, kept in (((1024, 1024), (256, 513)),     # Nyquist column kept
                          ((2048, 1024), (64, 33)),       # line length without a codelet
                          ((1024, 288), (64, 33)),        # 288 = 32 x 9: no radix-9 codelet
                          ((64, 64), (64, 33))):          # keep-everything: beyond the pruned 32-point stage's range
        plan = lib.plan_create(list(spatial), list(kept), fft_norm="forward", flags=0)
        try:
            assert lib.plan_kernel_name(plan, 0) != "k_f2p_r2c"
        finally:
            lib.plan_destroy(plan)


def test_two_pass_chunking_and_bias_offsets(tmp_path):
    """The host runs the two passes over chunks of images (192 MB of panel in the product); a build with a 1 MB chunk
    makes 8 images of 512 x 512 take two chunks here: the second chunk's bias index starts at (first image of the
    chunk) mod channels, and the results equal the one-chunk run of the product-sized build."""
    import os
    import subprocess
    from engine_runner import EMU_DIR, ROOT
    out = os.path.join(str(tmp_path), "libsc_engine_emu_chunk1.so")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-pthread", "-DSC_EMU=1", "-DSC_F2P_CHUNK_MB=1",
                           "-x", "c++", os.path.join(ROOT, "neuraloperator_amd", "csrc", "sc_engine.cpp"),
                           os.path.join(EMU_DIR, "sc_emu_runtime.cpp"), "-o", out])
    small = _lib.ScEngineLib(out)
    big = emu_lib()
    rng = np.random.default_rng(8)
    spatial, kept, n_img, channels = (512, 512), (16, 33), 8, 4
    x = torch.from_numpy(rng.standard_normal((n_img, *spatial)).astype(np.float32))
    yhat = torch.from_numpy((rng.standard_normal((n_img, *kept)) + 1j * rng.standard_normal((n_img, *kept))).astype(np.complex64))
    bias = torch.from_numpy(rng.standard_normal(channels).astype(np.float32))
    res = []
    for lib in (small, big):
        plan = lib.plan_create(list(spatial), list(kept), fft_norm="forward", flags=0)
        try:
            nbytes = lib.plan_workspace_bytes(plan, n_img)
            ws = torch.empty(max(nbytes, 256), dtype=torch.uint8)
            xhat = torch.empty((n_img, *kept), dtype=torch.complex64)
            lib.transform_forward(plan, _lib.SC_FWD_SCALED, x.data_ptr(), torch.view_as_real(xhat).data_ptr(), n_img, ws.data_ptr(), 0)
            y = torch.empty((n_img, *spatial), dtype=torch.float32)
            lib.transform_inverse(plan, _lib.SC_INV_PADDED, torch.view_as_real(yhat).data_ptr(), bias.data_ptr(), channels,
                                  y.data_ptr(), n_img, ws.data_ptr(), 0)
            res.append((xhat.numpy().copy(), y.numpy().copy(), nbytes))
        finally:
            lib.plan_destroy(plan)
    assert res[0][2] < res[1][2], "the small-chunk build asks for a smaller panel workspace"
    assert np.array_equal(res[0][0], res[1][0]) and np.array_equal(res[0][1], res[1][1])


def test_column_kernel_xcd_block_map():
    """k_f2p_col_inv_w1024 with a grid that is a multiple of 8 (the real chip: 512 persistent workgroups): XCD x = b % 8 walks
    the blocks [x ceil(n / 8), (x + 1) ceil(n / 8)) -- the column blocks of one image stay on one XCD.  The emulated chip has
    one compute unit (grid 2: the plain block order), so the XCD order runs in a subprocess with SC_F2P_COLW_WGS=8: 3 images =
    51 blocks, 7 per XCD, the last XCD's tail past the end."""
    import os
    import subprocess
    import sys
    code = r'''
import sys, numpy as np, torch
from engine_runner import emu_lib, rel_l2
from neuraloperator_amd import _lib
from test_emu_fft2p import _ref_inverse
lib = emu_lib()
rng = np.random.default_rng(4)
for kept in ((256, 129), (77, 20)):
    n_img, spatial = 3, (1024, 1024)
    plan = lib.plan_create(list(spatial), list(kept), fft_norm="forward", flags=0)
    ws = torch.empty(max(lib.plan_workspace_bytes(plan, n_img), 256), dtype=torch.uint8)
    yh = (rng.standard_normal((n_img, *kept)) + 1j * rng.standard_normal((n_img, *kept))).astype(np.complex64)
    y = torch.full((n_img, *spatial), float("nan"))
    lib.transform_inverse(plan, _lib.SC_INV_PADDED, torch.view_as_real(torch.from_numpy(yh)).data_ptr(), 0, n_img,
                          y.data_ptr(), n_img, ws.data_ptr(), 0)
    err = rel_l2(y.numpy(), _ref_inverse(yh, spatial, 1.0, True))
    print("ERR", kept, err)
    assert err < 2e-6, err
    lib.plan_destroy(plan)
'''
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, SC_F2P_COLW_WGS="8", PYTHONPATH=root + os.pathsep + os.path.join(root, "tests"))
    out = subprocess.run([sys.executable, "-c", code], cwd=root, env=env, capture_output=True, text=True, timeout=1500)
    assert out.returncode == 0 and out.stdout.count("ERR") == 2, out.stdout[-2000:] + out.stderr[-2000:]


@pytest.mark.parametrize("kept,n_img,mode", [((101, 129), 2, "padded"), ((256, 129), 5, "adjoint")])
def test_row_pass_adds_the_epilogue_skip_in_its_stores(lib, kept, n_img, mode):
    """Round 6: at 1024-point rows with 129 kept columns (configs[4]) the addend of sc_transform_inverse_ex (SC_ACT_NONE: the
    block backward's gradient around the spectral convolution, sc_layer_backward_ex) rides in k_f2p_c2r_w1024<true>'s store
    path instead of a streaming k_epilogue pass behind the transform: same two additions in the same order -> the same bits
    as the plain transform followed by the sum; other shapes keep the separate pass."""
    rng = np.random.default_rng(11)
    spatial = (1024, 1024)
    md = _lib.SC_INV_PADDED if mode == "padded" else _lib.SC_INV_ADJ_R2C
    plan = lib.plan_create(list(spatial), list(kept), fft_norm="forward", flags=0)
    try:
        assert lib.plan_kernel_name(plan, 0) == "k_f2p_r2c"
        ws = torch.empty(max(lib.plan_workspace_bytes(plan, n_img), 256), dtype=torch.uint8)
        yhat = torch.from_numpy((rng.standard_normal((n_img, *kept)) + 1j * rng.standard_normal((n_img, *kept))).astype(np.complex64))
        bias = torch.from_numpy(rng.standard_normal(n_img).astype(np.float32)) if mode == "padded" else None
        bp = 0 if bias is None else bias.data_ptr()
        skip = torch.from_numpy(rng.standard_normal((n_img, *spatial)).astype(np.float32))
        plain = torch.full((n_img, *spatial), float("nan"), dtype=torch.float32)
        lib.transform_inverse(plan, md, torch.view_as_real(yhat).data_ptr(), bp, n_img, plain.data_ptr(), n_img, ws.data_ptr(), 0)
        fused = torch.full((n_img, *spatial), float("nan"), dtype=torch.float32)
        lib.transform_inverse_ex(plan, md, torch.view_as_real(yhat).data_ptr(), bp, n_img, skip.data_ptr(), 0, _lib.SC_ACT_NONE,
                                 fused.data_ptr(), n_img, ws.data_ptr(), 0)
        assert torch.equal(fused, plain + skip)
    finally:
        lib.plan_destroy(plan)
