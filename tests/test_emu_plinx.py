"""CPU tier: the 1 x 1 maps with the block's pointwise operations in their load / store paths (csrc/sc_kernels_plinx.h, round 6)
in host emulation against the same computation in torch (float64 + autograd): every channel pair that needs its own
structure (rectangular, 128 -> 128 = two launches for the weight gradient), every option alone and all together, and the
two-pass form of the ChannelMLP + soft-gating skip + GELUs (channel_mlp.py:82-119, skip_connections.py:53-130,
fno_block.py:392-414) that blocks.PointwiseMLP2Fn / FusedBlockFn run for channel counts without a one-pass kernel."""
import pytest
import torch
import torch.nn.functional as F

from engine_runner import emu_lib, rel_l2
from neuraloperator_amd import _lib

TOL = 3e-6
XACT, ACT, PRO, XGRAD = _lib.SC_PLX_XACT, _lib.SC_PLX_ACT, _lib.SC_PLX_PRO, _lib.SC_PLX_XGRAD


@pytest.fixture(scope="module")
def lib():
    return emu_lib()


def p(t):
    return 0 if t is None else t.data_ptr()


def _fwd_ref(x, w, b, skip, gate, flags):
    xin = F.gelu(x) if flags & XACT else x
    z = torch.einsum("oc,bcs->bos", w, xin) + (0 if b is None else b[None, :, None])
    if skip is not None:
        z = z + gate[None, :, None] * skip
    return (F.gelu(z) if flags & ACT else z), z


@pytest.mark.parametrize("ci,co", [(32, 64), (128, 64), (64, 128), (128, 128), (128, 32)])
@pytest.mark.parametrize("flags,gated,bias", [(0, False, True), (XACT | ACT, True, True), (ACT, True, False), (XACT, False, False)])
def test_forward(lib, ci, co, flags, gated, bias):
    g = torch.Generator().manual_seed(ci + 3 * co + flags)
    B, S = 2, 96
    x = torch.randn(B, ci, S, generator=g)
    w = torch.randn(co, ci, generator=g) / ci ** 0.5
    b = torch.randn(co, generator=g) if bias else None
    skip = torch.randn(B, co, S, generator=g) if gated else None
    gate = torch.randn(co, generator=g) if gated else None
    out, pre = torch.full((B, co, S), float("nan")), torch.full((B, co, S), float("nan"))
    lib.pointwise_linear_forward_ex(B, ci, co, S, flags, p(x), p(w), p(b), p(skip), p(gate), p(out), p(pre), 0)
    d = lambda t: None if t is None else t.double()
    ro, rz = _fwd_ref(d(x), d(w), d(b), d(skip), d(gate), flags)
    assert rel_l2(out.numpy(), ro.numpy()) < TOL and rel_l2(pre.numpy(), rz.numpy()) < TOL
    out2 = torch.full((B, co, S), float("nan"))
    lib.pointwise_linear_forward_ex(B, ci, co, S, flags, p(x), p(w), p(b), p(skip), p(gate), p(out2), 0, 0)      # no pre_out
    assert torch.equal(out2, out)


@pytest.mark.parametrize("ci,co", [(32, 64), (128, 64), (64, 128), (128, 128), (64, 32)])
@pytest.mark.parametrize("flags,gated,addend", [(0, False, False), (XACT | XGRAD | PRO, True, False), (XGRAD, False, True),
                                                (PRO, True, True)])
def test_backward(lib, ci, co, flags, gated, addend):
    """out = act(W xin + gate (.) skip), loss = <out, gout> (+ <x-branch, addend>): every gradient of the pass against autograd
    in float64.  PRO: the forward output went through a GELU (pre = its input); XACT: xin = gelu(x); XGRAD: gx is taken
    through gelu'(xg) -- for an XACT layer xg = x (gradient of the pre-activation), otherwise an independent tensor (the
    Fourier layer's pre-activation of the block)."""
    g = torch.Generator().manual_seed(ci + 5 * co + flags)
    B, S = 2, 96
    x = torch.randn(B, ci, S, generator=g)
    w = torch.randn(co, ci, generator=g) / ci ** 0.5
    gout = torch.randn(B, co, S, generator=g)
    skip = torch.randn(B, co, S, generator=g) if gated else None
    gate = torch.randn(co, generator=g) if gated else None
    add = torch.randn(B, ci, S, generator=g) if addend else None
    xg = x if flags & XACT else (torch.randn(B, ci, S, generator=g) if flags & XGRAD else None)
    # float64 reference
    xd, wd = x.double().requires_grad_(True), w.double().requires_grad_(True)
    sd = None if skip is None else skip.double().requires_grad_(True)
    gd = None if gate is None else gate.double().requires_grad_(True)
    xin = F.gelu(xd) if flags & XACT else xd
    z = torch.einsum("oc,bcs->bos", wd, xin)
    if gated:
        z = z + gd[None, :, None] * sd
    out = F.gelu(z) if flags & PRO else z
    out.backward(gout.double())
    pre = z.detach().float().contiguous() if flags & PRO else None      # (einsum may hand back a permuted view)
    gx_ref = xd.grad
    if flags & XGRAD and not flags & XACT:                 # an independent xg: (W^T g + addend) (.) gelu'(xg)
        t = xg.double().requires_grad_(True)
        F.gelu(t).backward(torch.ones_like(t))
        gx_ref = (gx_ref + (0 if add is None else add.double())) * t.grad
    elif flags & XACT:                                     # autograd already went through gelu(x); an addend joins before it
        if add is not None:
            t = x.double().requires_grad_(True)
            F.gelu(t).backward(add.double())
            gx_ref = gx_ref + t.grad
    elif add is not None:
        gx_ref = gx_ref + add.double()
    # engine
    gx, gw, gb = torch.full_like(x, float("nan")), torch.full_like(w, float("nan")), torch.full((co,), float("nan"))
    gsk = torch.full((B, co, S), float("nan")) if gated else None
    ggt = torch.full((co,), float("nan")) if gated else None
    ws = torch.empty(lib.pointwise_linear_workspace_bytes_ex(B, ci, co, S), dtype=torch.uint8)
    lib.pointwise_linear_backward_ex(B, ci, co, S, flags, p(x), p(w), p(gout), p(pre), p(xg), p(skip), p(gate), p(add), p(gx), p(gw),
                                     p(gb), p(gsk), p(ggt), p(ws), 0)
    assert rel_l2(gx.numpy(), gx_ref.numpy()) < TOL
    assert rel_l2(gw.numpy(), wd.grad.numpy()) < TOL
    gz = gout.double()
    if flags & PRO:
        t = z.detach().requires_grad_(True)
        F.gelu(t).backward(gout.double())
        gz = t.grad
    assert rel_l2(gb.numpy(), gz.sum((0, 2)).numpy()) < TOL
    if gated:
        assert rel_l2(gsk.numpy(), sd.grad.numpy()) < TOL and rel_l2(ggt.numpy(), gd.grad.numpy()) < TOL
    # the weight gradients alone (gx = NULL): same bits -- except for the plain map (no option, no gate), which takes the LEAN
    # instantiation on twice the workgroups (another order of the partial sums)
    gw2, gb2 = torch.full_like(w, float("nan")), torch.full((co,), float("nan"))
    ggt2 = torch.full((co,), float("nan")) if gated else None
    lib.pointwise_linear_backward_ex(B, ci, co, S, flags & ~XGRAD, p(x), p(w), p(gout), p(pre), 0, p(skip), p(gate), 0, 0, p(gw2),
                                     p(gb2), 0, p(ggt2), p(ws), 0)
    if (flags & ~XGRAD) == 0 and not gated:
        assert rel_l2(gw2.numpy(), gw.numpy()) < 1e-6 and rel_l2(gb2.numpy(), gb.numpy()) < 1e-6
    else:
        assert torch.equal(gw2, gw) and torch.equal(gb2, gb) and (not gated or torch.equal(ggt2, ggt))


@pytest.mark.parametrize("chans", [(128, 64, 128), (128, 128, 128), (64, 128, 64)], ids=str)
@pytest.mark.parametrize("act", [1, 0])
def test_two_pass_channel_mlp_with_gated_skip(lib, chans, act):
    """fc1, fc2 as two passes each way (what blocks.PointwiseMLP2Fn issues) == autograd of
    act(W2 gelu(W1 x + b1) + b2 + gate (.) skip) in float64"""
    ci, ch, co = chans
    g = torch.Generator().manual_seed(ci + ch + act)
    B, S = 2, 64
    x = torch.randn(B, ci, S, generator=g)
    w1, b1 = torch.randn(ch, ci, generator=g) / ci ** 0.5, torch.randn(ch, generator=g)
    w2, b2 = torch.randn(co, ch, generator=g) / ch ** 0.5, torch.randn(co, generator=g)
    skip, gate, gout = torch.randn(B, co, S, generator=g), torch.randn(co, generator=g), torch.randn(B, co, S, generator=g)
    leaves = [t.double().requires_grad_(True) for t in (x, w1, b1, w2, b2, skip, gate)]
    xd, w1d, b1d, w2d, b2d, sd, gd = leaves
    h = F.gelu(torch.einsum("hc,bcs->bhs", w1d, xd) + b1d[None, :, None])
    z = torch.einsum("oh,bhs->bos", w2d, h) + b2d[None, :, None] + gd[None, :, None] * sd
    ref = F.gelu(z) if act else z
    ref.backward(gout.double())
    hpre, out = torch.empty(B, ch, S), torch.empty(B, co, S)
    zpre = torch.empty(B, co, S) if act else None
    lib.pointwise_linear_forward_ex(B, ci, ch, S, 0, p(x), p(w1), p(b1), 0, 0, p(hpre), 0, 0)
    lib.pointwise_linear_forward_ex(B, ch, co, S, XACT | (ACT if act else 0), p(hpre), p(w2), p(b2), p(skip), p(gate), p(out), p(zpre), 0)
    assert rel_l2(out.numpy(), ref.detach().numpy()) < TOL
    ghp, gx = torch.empty_like(hpre), torch.empty_like(x)
    gw1, gw2, gb1, gb2 = torch.empty_like(w1), torch.empty_like(w2), torch.empty_like(b1), torch.empty_like(b2)
    gsk, ggt = torch.empty_like(skip), torch.empty_like(gate)
    ws = torch.empty(max(lib.pointwise_linear_workspace_bytes_ex(B, ch, co, S), lib.pointwise_linear_workspace_bytes_ex(B, ci, ch, S)),
                     dtype=torch.uint8)
    lib.pointwise_linear_backward_ex(B, ch, co, S, XACT | XGRAD | (PRO if act else 0), p(hpre), p(w2), p(gout), p(zpre), p(hpre),
                                     p(skip), p(gate), 0, p(ghp), p(gw2), p(gb2), p(gsk), p(ggt), p(ws), 0)
    lib.pointwise_linear_backward_ex(B, ci, ch, S, 0, p(x), p(w1), p(ghp), 0, 0, 0, 0, 0, p(gx), p(gw1), p(gb1), 0, 0, p(ws), 0)
    for got, want in zip((gx, gw1, gb1, gw2, gb2, gsk, ggt), leaves):
        assert rel_l2(got.numpy(), want.grad.numpy()) < TOL


def test_argument_checks(lib):
    x, w, out = torch.zeros(1, 64, 32), torch.zeros(128, 64), torch.zeros(1, 128, 32)
    with pytest.raises(_lib.EngineError):                # 48 channels
        lib.pointwise_linear_forward_ex(1, 48, 128, 32, 0, p(x), p(w), 0, 0, 0, p(out), 0, 0)
    with pytest.raises(_lib.EngineError):                # spatial not a multiple of 32
        lib.pointwise_linear_forward_ex(1, 64, 128, 40, 0, p(x), p(w), 0, 0, 0, p(out), 0, 0)
    with pytest.raises(_lib.EngineError):                # 2^28 points per image: lane offsets are 32-bit byte counts (no launch)
        lib.pointwise_linear_forward_ex(1, 64, 128, 1 << 28, 0, p(x), p(w), 0, 0, 0, p(out), 0, 0)
    with pytest.raises(_lib.EngineError):                # a gate without its source
        lib.pointwise_linear_forward_ex(1, 64, 128, 32, 0, p(x), p(w), 0, 0, p(w), p(out), 0, 0)
    with pytest.raises(_lib.EngineError):                # a backward flag in a forward call
        lib.pointwise_linear_forward_ex(1, 64, 128, 32, PRO, p(x), p(w), 0, 0, 0, p(out), 0, 0)
    ws = torch.empty(lib.pointwise_linear_workspace_bytes_ex(1, 64, 128, 32), dtype=torch.uint8)
    with pytest.raises(_lib.EngineError):                # PRO without the pre-activation
        lib.pointwise_linear_backward_ex(1, 64, 128, 32, PRO, p(x), p(w), p(out), 0, 0, 0, 0, 0, p(x), p(w), 0, 0, 0, p(ws), 0)
    assert lib.pointwise_linear_workspace_bytes_ex(1, 48, 64, 32) == 0
