"""GPU tier (``-m gpu``): oracle compares AT the BASELINE.json configurations (VERDICT r1 "what's weak" 1).

Every case runs the whole layer forward + backward through the C-ABI on the MI355X and compares y, gx, gW and
gbias with the CPU oracle (``oracle.spectral_oracle.forward_torch`` + autograd = the reference's
spectral_convolution.py:417-570 and its implicit backward) on the same seeded inputs.  Bar: rel-L2 <= 1e-5 in
fp32 (north star).  Sizes:

  C2  headline   B=32, C=64, 256^2, modes (64,64)          -- the metric shape at its full batch
  C4  FNO3d      B=8,  C=32, 128^3, modes (32,32,32)       -- configs[3] at its bench batch (B = 2, the share at 4 ranks, until round 4)
  C5  FNO2d      B=1,  16 -> 128 channels, 1024^2, modes (256,256) -- the large-grid passes at N = 1024,
                 J = 129, K = 256 and the hidden-128 contraction (Q = 128)
  C3  TFNO       B=4,  C=64, 256^2, Tucker rank 0.1 -> (36,36,36,19), factorized and reconstructed,
                 gradients of the core and of every factor

Round 3 (VERDICT r2 "what's weak" 1):
  C4  at its BENCH batch B=8 (small-batch streamed route + backward pair launch + k_pl128_* + k_ax128 as a layer)
  C2  bf16 real-tensor I/O at the literal (32, 64, 64, 256^2, modes 64) shape
  C2  with STRUCTURED inputs -- x = 1e3 + randn (DC-dominated spectrum), a smooth low-frequency field, a 0/1
      Darcy-like coefficient field lifted to C = 64 -- so the three-real-product contraction (errors scale with
      |A||B|, not with |Re| and |Im| separately) is judged on non-Gaussian spectra; distance to the float64
      restatement (forward_np64 / backward_np64) is reported next to the distance to the fp32 reference path.

Round 4 (VERDICT r3 "what's weak" 1):
  C5  LITERALLY: B = 4, 128 -> 128 channels, 1024^2, modes (256, 256) (test_c5_literal_shape_vs_oracle)
  C3  also at the bench batch B = 32
"""
import numpy as np
import pytest
import torch

from engine_runner import layer_fwd_bwd, rel_l2

pytestmark = pytest.mark.gpu
TOL = 1e-5


@pytest.fixture(scope="module")
def lib():
    from neuraloperator_amd import _lib
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    return _lib.get_lib()


AT_CONFIG = [
    ("C2_fno2d_256_m64_c64_b32", 32, 64, 64, (256, 256), (64, 64)),
    # (C4 at B = 2 -- the per-GPU share at 4 ranks -- was here until round 4: the B = 8 case below runs the same kernels)
    ("C4_fno3d_128_m32_c32_b8", 8, 32, 32, (128, 128, 128), (32, 32, 32)),      # BASELINE configs[3] at its bench batch
    ("C5_fno2d_1024_m256_c16to128_b1", 1, 16, 128, (1024, 1024), (256, 256)),
    # (C5 at 128 -> 16 channels, B = 2, was here in round 3: superseded by test_c5_literal_shape_vs_oracle -- the literal
    #  128 -> 128, B = 4 shape -- and test_backward_pair_small_batch_one_pass, which runs every batch size of the one-pass
    #  backward kernel; dropped to keep the GPU tier near ten minutes)
]


@pytest.mark.parametrize("case", AT_CONFIG, ids=lambda c: c[0])
def test_layer_vs_oracle_at_config(lib, case):
    from oracle import spectral_oracle as so
    from neuraloperator_amd.modes import halve_last_mode

    _, b, ci, co, spatial, modes = case
    torch.manual_seed(4321)
    nm = halve_last_mode(modes)
    std = (2 / (ci + co)) ** 0.5
    x = torch.randn(b, ci, *spatial)
    w = torch.empty(ci, co, *nm, dtype=torch.cfloat).normal_(0, std)
    bias = std * torch.randn(co, *(1,) * len(spatial))
    g = torch.randn(b, co, *spatial)
    dev = torch.device("cuda:0")
    y, gx, gw, gb, xh = layer_fwd_bwd(lib, x.to(dev), w.to(dev), bias.to(dev), g.to(dev), nm, nm)
    y, gx, gw, gb = y.cpu().numpy(), gx.cpu().numpy(), gw.cpu().numpy(), gb.cpu().numpy()
    torch.cuda.empty_cache()
    xc, wc, bc = x.requires_grad_(True), w.requires_grad_(True), bias.requires_grad_(True)
    yo = so.forward_torch(xc, wc, bc, nm, nm)
    yo.backward(g)
    errs = dict(y=rel_l2(y, yo.detach().numpy()), gx=rel_l2(gx, xc.grad.numpy()),
                gw=rel_l2(gw, wc.grad.numpy()), gb=rel_l2(gb, bc.grad.numpy()))
    print(case[0], " ".join(f"{k}={v:.2e}" for k, v in errs.items()))
    assert all(np.isfinite(v) and v < TOL for v in errs.values()), errs


def test_c5_literal_shape_vs_oracle(lib):
    """BASELINE configs[4] LITERALLY: B = 4, 128 -> 128 channels, 1024^2, modes (256, 256) --
    the two-pass P = 32 transforms, k_modegemm_sb forward and the one-pass backward pair k_modegemm_sb_bwd<4,4,2> over
    the 4.33 GB weight, exactly as `extra.fno2d_1024_b4` times them.  The oracle (forward_torch + autograd =
    spectral_convolution.py:417-570) runs one sample at a time to bound host memory: y[b] and gx[b] depend on sample b
    only, gW and gbias are sums over the samples (accumulated in float64 on the host)."""
    from oracle import spectral_oracle as so
    from neuraloperator_amd.modes import halve_last_mode

    b, ci, co, spatial, modes = 4, 128, 128, (1024, 1024), (256, 256)
    torch.manual_seed(1234)
    nm = halve_last_mode(modes)
    std = (2 / (ci + co)) ** 0.5
    x = torch.randn(b, ci, *spatial)
    w = torch.empty(ci, co, *nm, dtype=torch.cfloat).normal_(0, std)
    bias = std * torch.randn(co, 1, 1)
    g = torch.randn(b, co, *spatial)
    dev = torch.device("cuda:0")
    y, gx, gw, gb, _ = layer_fwd_bwd(lib, x.to(dev), w.to(dev), bias.to(dev), g.to(dev), nm, nm)
    y, gx, gw, gb = y.cpu(), gx.cpu(), gw.cpu(), gb.cpu()
    torch.cuda.empty_cache()
    gw_ref = torch.zeros(ci, co, *nm, dtype=torch.complex128)
    gb_ref = torch.zeros(co, dtype=torch.float64)
    num = dict(y=0.0, gx=0.0)
    den = dict(y=0.0, gx=0.0)
    step = 1          # sample by sample: one oracle call for the whole batch is 2.5 x SLOWER on the GPU box's host (206 vs 80 s)
    for s in range(0, b, step):
        xc = x[s:s + step].clone().requires_grad_(True)
        wc = w.clone().requires_grad_(True)
        bc = bias.clone().requires_grad_(True)
        yo = so.forward_torch(xc, wc, bc, nm, nm)
        yo.backward(g[s:s + step])
        num["y"] += float((y[s:s + step].double() - yo.detach().double()).pow(2).sum())
        den["y"] += float(yo.detach().double().pow(2).sum())
        num["gx"] += float((gx[s:s + step].double() - xc.grad.double()).pow(2).sum())
        den["gx"] += float(xc.grad.double().pow(2).sum())
        gw_ref += wc.grad
        gb_ref += bc.grad.reshape(-1).double()
        del xc, wc, bc, yo
    errs = dict(y=(num["y"] / den["y"]) ** 0.5, gx=(num["gx"] / den["gx"]) ** 0.5,
                gw=rel_l2(gw.numpy(), gw_ref.numpy()),
                gb=rel_l2(gb.numpy().reshape(-1), gb_ref.numpy()))
    print("C5 literal", " ".join(f"{k}={v:.2e}" for k, v in errs.items()))
    assert all(np.isfinite(v) and v < TOL for v in errs.values()), errs


@pytest.mark.parametrize("b", [4, 32], ids=["B4", "B32_bench_batch"])
@pytest.mark.parametrize("impl", ["factorized", "factorized_fused_chain", "reconstructed"])
def test_tfno_tucker_rank01_at_config(impl, b, monkeypatch):
    """BASELINE configs[2]: TFNO2d Tucker rank 0.1 at C=64, 256^2, modes (64,64) -> ranks (36,36,36,19), through
    the drop-in module; reference = the oracle's pairwise contraction (SURVEY 8 row a6 order) with autograd.
    B = 32 is the batch `extra.tfno_rank01` of the bench line runs (the per-mode products take other kernel routes
    than at B = 4)."""
    from oracle import spectral_oracle as so
    from neuraloperator_amd import SpectralConv

    dev = torch.device("cuda:0")
    torch.manual_seed(99)
    c, n = 64, 256
    if impl == "factorized_fused_chain":                  # round 5: the opt-in one-launch-each-way chain through the module
        monkeypatch.setenv("SC_TKC", "1")
        impl = "factorized"
    conv = SpectralConv(c, c, (64, 64), factorization="Tucker", rank=0.1, implementation=impl).to(dev)
    assert tuple(conv.weight.core.shape) == (36, 36, 36, 19)
    with torch.no_grad():
        conv.weight.core.copy_(torch.randn(36, 36, 36, 19, dtype=torch.cfloat) * 0.3)
        for f in conv.weight.factors:
            f.copy_(torch.randn(*f.shape, dtype=torch.cfloat) * 0.3)
    x = torch.randn(b, c, n, n)
    g = torch.randn(b, c, n, n)
    xd = x.to(dev).requires_grad_(True)
    y = conv(xd)
    y.backward(g.to(dev))
    torch.cuda.synchronize()
    core = conv.weight.core.detach().cpu().requires_grad_(True)
    facs = [f.detach().cpu().requires_grad_(True) for f in conv.weight.factors]
    bias = conv.bias.detach().cpu().requires_grad_(True)
    xc = x.requires_grad_(True)
    nm = list(conv.n_modes)
    contract = lambda xk, wk: so.contract_tucker(xk, core, facs)
    yo = so.forward_torch(xc, so.reconstruct_tucker(core, facs).detach(), bias, nm, nm, contract=contract)
    yo.backward(g)
    errs = dict(y=rel_l2(y.detach().cpu().numpy(), yo.detach().numpy()),
                gx=rel_l2(xd.grad.cpu().numpy(), xc.grad.numpy()),
                gb=rel_l2(conv.bias.grad.cpu().numpy(), bias.grad.numpy()),
                g_core=rel_l2(conv.weight.core.grad.cpu().numpy(), core.grad.numpy()))
    for i, f in enumerate(facs):
        errs[f"g_factor_{i}"] = rel_l2(conv.weight.factors[i].grad.cpu().numpy(), f.grad.numpy())
    print(impl, " ".join(f"{k}={v:.2e}" for k, v in errs.items()))
    assert all(np.isfinite(v) and v < TOL for v in errs.values()), errs


@pytest.mark.parametrize("dims", [(32, 64, 64, 36, 36, 2112), (4, 64, 64, 36, 36, 2112), (16, 32, 48, 20, 28, 1000)],
                         ids=lambda d: "B%d_Ci%d_Co%d_R%d_%d_M%d" % d)
def test_fused_tucker_chain_on_device(dims, monkeypatch):
    """The fused chain kernels (csrc/sc_kernels_tkchain.h) through the C-ABI at configs[2]'s literal extents (persistent
    workgroups over 528 four-mode tiles, three rounds on some units) against a complex128 einsum on the device: z, t,
    yhat and all four gradients; two backward calls give identical bits (fixed-order reduction of the factor
    gradients); the nine launches of rounds 3-4 agree to round-off."""
    from neuraloperator_amd import _lib
    lib = _lib.get_lib()
    dev = torch.device("cuda:0")
    B, Ci, Co, R1, R2, M = dims
    monkeypatch.setenv("SC_TKC", "1")                     # opt-in path (slower than the nine launches: DESIGN 8)
    assert lib.tucker_chain_fused_supported(dims)
    torch.manual_seed(5)
    rnd = lambda *sh: torch.randn(*sh, dtype=torch.complex64, device=dev)
    xhat, u_in, t3, u_out, gy = rnd(B, Ci, M), rnd(Ci, R1), rnd(R1, R2, M), rnd(Co, R2), rnd(B, Co, M)
    new = lambda *sh: torch.full(sh, float("nan"), dtype=torch.complex64, device=dev)
    p = lambda v: v.data_ptr()
    st = torch.cuda.current_stream().cuda_stream
    z, t, yhat, t3m = new(B, R1, M), new(B, R2, M), new(B, Co, M), new(M, R1, R2)
    lib.tucker_chain_forward_fused(dims, p(xhat), p(u_in), p(t3), p(u_out), p(t3m), p(z), p(t), p(yhat), st)
    assert torch.equal(torch.view_as_real(t3m), torch.view_as_real(t3.permute(2, 0, 1).contiguous()))
    c = lambda v: v.to(torch.complex128)
    Z = torch.einsum("bim,if->bfm", c(xhat), c(u_in))
    T = torch.einsum("bfm,fgm->bgm", Z, c(t3))
    Y = torch.einsum("bgm,og->bom", T, c(u_out))
    rel = lambda a, b: float((c(a) - b).norm() / b.norm())
    errs = dict(z=rel(z, Z), t=rel(t, T), yhat=rel(yhat, Y))
    gT = torch.einsum("bom,og->bgm", c(gy), c(u_out).conj())
    gZ = torch.einsum("bgm,fgm->bfm", gT, c(t3).conj())
    ref = dict(gx=torch.einsum("bfm,if->bim", gZ, c(u_in).conj()), gt3=torch.einsum("bfm,bgm->fgm", Z.conj(), gT),
               gu_in=torch.einsum("bim,bfm->if", c(xhat).conj(), gZ), gu_out=torch.einsum("bgm,bom->og", T.conj(), c(gy)))
    nb = lib.tucker_chain_backward_fused_workspace_bytes(dims)
    ws = torch.empty(nb, dtype=torch.uint8, device=dev)
    runs = []
    for _ in range(2):
        got = dict(gx=new(B, Ci, M), gu_in=new(Ci, R1), gt3=new(R1, R2, M), gu_out=new(Co, R2))
        lib.tucker_chain_backward_fused(dims, p(xhat), p(u_in), p(t3m), p(u_out), p(z), p(t), p(gy), p(got["gx"]), p(got["gu_in"]),
                                        p(got["gt3"]), p(got["gu_out"]), ws.data_ptr(), nb, st)
        runs.append(got)
    torch.cuda.synchronize()
    for k in ref:
        errs[k] = rel(runs[0][k], ref[k])
        assert torch.equal(torch.view_as_real(runs[0][k]), torch.view_as_real(runs[1][k])), k
    z9, t9, y9 = new(B, R1, M), new(B, R2, M), new(B, Co, M)
    lib.tucker_chain_forward(dims, p(xhat), p(u_in), p(t3), p(u_out), p(z9), p(t9), p(y9), st)
    errs["yhat_vs_nine_launches"] = rel(yhat, c(y9))
    print(dims, " ".join(f"{k}={v:.1e}" for k, v in errs.items()))
    assert all(np.isfinite(v) and v < 3e-6 for v in errs.values()), errs


def _structured(kind, b, c, n, gen):
    """Non-Gaussian inputs at the metric shape."""
    if kind == "dc_offset":                               # |DC| = 1e3 against O(1) everywhere else
        return 1e3 + torch.randn(b, c, n, n, generator=gen)
    t = torch.arange(n, dtype=torch.float32) / n
    yy, xx = torch.meshgrid(t, t, indexing="ij")
    if kind == "smooth":                                  # a few low modes with 1/k^2 amplitudes + a small rough part
        f = torch.zeros(b, c, n, n)
        for k1 in range(0, 5):
            for k2 in range(0, 5):
                a = torch.randn(b, c, 1, 1, generator=gen) / (1.0 + k1 * k1 + k2 * k2)
                ph = 6.2831853 * torch.rand(b, c, 1, 1, generator=gen)
                f = f + a * torch.cos(6.2831853 * (k1 * yy + k2 * xx) + ph)
        return f + 1e-3 * torch.randn(b, c, n, n, generator=gen)
    if kind == "darcy01":                                 # thresholded smooth field: piecewise-constant 0 / 1 coefficient,
        base = torch.zeros(b, 1, n, n)                    # lifted to C channels by a random 1 x 1 map (as FNO's lifting does)
        for k1 in range(1, 4):
            for k2 in range(1, 4):
                a = torch.randn(b, 1, 1, 1, generator=gen) / (k1 * k1 + k2 * k2)
                ph = 6.2831853 * torch.rand(b, 1, 1, 1, generator=gen)
                base = base + a * torch.sin(6.2831853 * (k1 * yy + k2 * xx) + ph)
        field = (base > 0).float()
        lift = torch.randn(1, c, 1, 1, generator=gen)
        off = torch.randn(1, c, 1, 1, generator=gen)
        return field * lift + off
    raise ValueError(kind)


@pytest.mark.parametrize("kind", ["dc_offset", "smooth", "darcy01"])
def test_metric_shape_structured_inputs(lib, kind):
    """The metric shape (B = 32, C = 64, 256^2, modes 64: k_fft2d_*3 + the three-product k_modegemm_dma + the pair
    launch) on inputs whose spectra are NOT iid Gaussian.  Bars: rel-L2 <= 1e-5 against the fp32 reference path
    (spectral_convolution.py:417-570 via oracle.forward_torch + autograd) -- the north-star bar -- and <= 1e-5 against
    the float64 restatement, whose distance to the fp32 CPU path is printed beside it."""
    from oracle import spectral_oracle as so
    from neuraloperator_amd.modes import halve_last_mode

    b, c, n, modes = 32, 64, 256, (64, 64)
    gen = torch.Generator().manual_seed(77)
    nm = halve_last_mode(modes)
    std = (2 / (2 * c)) ** 0.5
    x = _structured(kind, b, c, n, gen).contiguous()
    w = torch.view_as_complex(torch.randn(c, c, *nm, 2, generator=gen) * (std / 2 ** 0.5))
    bias = std * torch.randn(c, 1, 1, generator=gen)
    g = _structured(kind, b, c, n, gen).contiguous()          # the upstream gradient is structured as well
    dev = torch.device("cuda:0")
    y, gx, gw, gb, _ = layer_fwd_bwd(lib, x.to(dev), w.to(dev), bias.to(dev), g.to(dev), nm, nm)
    y, gx, gw, gb = y.cpu().numpy(), gx.cpu().numpy(), gw.cpu().numpy(), gb.cpu().numpy()
    torch.cuda.empty_cache()
    xc, wc, bc = x.clone().requires_grad_(True), w.clone().requires_grad_(True), bias.clone().requires_grad_(True)
    yo = so.forward_torch(xc, wc, bc, nm, nm)
    yo.backward(g)
    ref32 = dict(y=yo.detach().numpy(), gx=xc.grad.numpy(), gw=wc.grad.numpy(), gb=bc.grad.numpy())
    y64, _ = so.forward_np64(x.numpy(), w.numpy(), bias.numpy(), nm, nm)
    gx64, gw64, gb64 = so.backward_np64(x.numpy(), w.numpy(), g.numpy(), nm, nm)
    ref64 = dict(y=y64, gx=gx64, gw=gw64, gb=gb64.reshape(ref32["gb"].shape))
    got = dict(y=y, gx=gx, gw=gw, gb=gb.reshape(ref32["gb"].shape))
    e32 = {k: rel_l2(got[k], ref32[k]) for k in got}
    e64 = {k: rel_l2(got[k], ref64[k]) for k in got}
    r64 = {k: rel_l2(ref32[k], ref64[k]) for k in got}
    print(kind, "vs fp32 ref:", " ".join(f"{k}={v:.2e}" for k, v in e32.items()),
          "| vs fp64:", " ".join(f"{k}={v:.2e}" for k, v in e64.items()),
          "| fp32 ref vs fp64:", " ".join(f"{k}={v:.2e}" for k, v in r64.items()))
    assert all(np.isfinite(v) and v < TOL for v in e32.values()), e32
    assert all(np.isfinite(v) and v < TOL for v in e64.values()), e64


def test_bf16_io_at_metric_shape(lib):
    """BASELINE configs[1] literally: bf16 real-tensor I/O at (B = 32, C = 64 -> 64, 256^2, modes 64).  Bars as in
    test_bf16_io_vs_oracle: y / gx within one bf16 ulp of the fp32 oracle evaluated on the SAME bf16
    inputs and > 98 % bit-identical to the rounded oracle; gW / gbias (fp32 in HBM) at 1e-5."""
    from oracle import spectral_oracle as so
    from neuraloperator_amd import _lib as L
    from neuraloperator_amd.modes import halve_last_mode

    b, c, n, modes = 32, 64, 256, (64, 64)
    torch.manual_seed(2024)
    nm = halve_last_mode(modes)
    std = (2 / (2 * c)) ** 0.5
    x = torch.randn(b, c, n, n).bfloat16()
    w = torch.empty(c, c, *nm, dtype=torch.cfloat).normal_(0, std)
    bias = std * torch.randn(c, 1, 1)
    g = torch.randn(b, c, n, n).bfloat16()
    dev = torch.device("cuda:0")
    y, gx, gw, gb, _ = layer_fwd_bwd(lib, x.to(dev), w.to(dev), bias.to(dev), g.to(dev), nm, nm, flags=L.SC_PLAN_IO_BF16)
    assert y.dtype == torch.bfloat16 and gx.dtype == torch.bfloat16
    y, gx, gw, gb = y.cpu(), gx.cpu(), gw.cpu().numpy(), gb.cpu().numpy()
    torch.cuda.empty_cache()
    xc, wc, bc = x.float().requires_grad_(True), w.requires_grad_(True), bias.requires_grad_(True)
    yo = so.forward_torch(xc, wc, bc, nm, nm)
    yo.backward(g.float())
    from test_gpu_parity import _bf16_checks
    _bf16_checks(y, yo.detach(), "y")
    _bf16_checks(gx, xc.grad, "gx")
    assert rel_l2(gw, wc.grad.numpy()) < TOL
    assert rel_l2(gb.reshape(-1), bc.grad.numpy().reshape(-1)) < TOL
