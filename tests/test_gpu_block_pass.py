"""GPU tier: sc_pointwise_block_forward (k_pblock_fwd) on the device -- the whole pointwise side of a default FNO
block's forward in one pass (fno_block.py:392-412: linear skip, add, GELU, ChannelMLP, soft-gating skip, GELU) against
float64 torch, and its s / y outputs against the passes it replaces (bit for bit)."""
import pytest
import torch
import torch.nn.functional as F

from neuraloperator_amd import _lib
from engine_runner import rel_l2

pytestmark = pytest.mark.gpu
TOL = 3e-6


@pytest.fixture(scope="module")
def lib():
    if not torch.cuda.is_available():
        pytest.skip("GPU tier: no GPU visible")
    return _lib.get_lib()


@pytest.mark.parametrize("c,ch,B,S", [(64, 32, 32, 256 * 256), (64, 64, 3, 4096), (32, 32, 5, 96), (64, 32, 2, 32)], ids=str)
@pytest.mark.parametrize("act", [1, 0])
def test_block_pass_on_device(lib, c, ch, B, S, act):
    dev = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(c * 7 + ch + act)
    mk = lambda *sh: torch.randn(*sh, generator=g).to(dev)
    x, conv = mk(B, c, S), mk(B, c, S)
    ws, w1, w2 = mk(c, c) / c ** 0.5, mk(ch, c) / c ** 0.5, mk(c, ch) / ch ** 0.5
    bs, b1, b2, gt = mk(c), mk(ch), mk(c), mk(c)
    y, pre, out = (torch.full((B, c, S), float("nan"), device=dev) for _ in range(3))
    st = torch.cuda.current_stream().cuda_stream
    p = lambda t: t.data_ptr()
    lib.pointwise_block_forward(B, c, ch, S, act, p(conv), p(x), p(ws), p(bs), p(w1), p(b1), p(w2), p(b2), p(gt), p(y),
                                p(pre) if act else 0, p(out), st)
    # the passes it replaces: linear skip, add (+ GELU), MLP pass
    skip, out3 = torch.empty_like(x), torch.empty_like(x)
    lib.pointwise_linear_forward(B, c, c, S, p(x), p(ws), p(bs), p(skip), st)
    s3 = conv + skip
    torch.cuda.synchronize()
    assert torch.equal(pre if act else y, s3)
    lib.pointwise_mlp_forward(B, c, ch, c, S, act, p(y), p(w1), p(b1), p(w2), p(b2), p(x), p(gt), p(out3), st)
    torch.cuda.synchronize()
    assert rel_l2(out.cpu().numpy(), out3.cpu().numpy()) < TOL
    # float64 (a slice of the large case)
    nb = min(B, 2)
    xd, cd = x[:nb].double(), conv[:nb].double()
    sd = cd + torch.einsum("oc,bcs->bos", ws.double(), xd) + bs.double()[None, :, None]
    yd = F.gelu(sd) if act else sd
    hd = F.gelu(torch.einsum("hc,bcs->bhs", w1.double(), yd) + b1.double()[None, :, None])
    zd = torch.einsum("oh,bhs->bos", w2.double(), hd) + b2.double()[None, :, None] + gt.double()[None, :, None] * xd
    od = F.gelu(zd) if act else zd
    assert rel_l2(y[:nb].cpu().numpy(), yd.cpu().numpy()) < TOL
    assert rel_l2(out[:nb].cpu().numpy(), od.cpu().numpy()) < TOL


def test_block_pass_derivative_form_on_device(lib):
    """SC_ACT_GELU_DGRAD at the metric shape's channel counts: `pre` = gelu'(s), y / out unchanged, and the backward pass
    with that buffer as x_pre returns what it returns with the pre-activation (round 6)."""
    dev = torch.device("cuda:0")
    c, ch, B, S = 64, 32, 4, 4096
    g = torch.Generator(device="cpu").manual_seed(11)
    mk = lambda *sh: torch.randn(*sh, generator=g).to(dev)
    x, conv, gout = mk(B, c, S), mk(B, c, S), mk(B, c, S)
    ws, w1, w2 = mk(c, c) / c ** 0.5, mk(ch, c) / c ** 0.5, mk(c, ch) / ch ** 0.5
    bs, b1, b2, gt = mk(c), mk(ch), mk(c), mk(c)
    st = torch.cuda.current_stream().cuda_stream
    p = lambda t: t.data_ptr()
    res = {}
    for act in (_lib.SC_ACT_GELU, _lib.SC_ACT_GELU_DGRAD):
        y, pre, out = (torch.full((B, c, S), float("nan"), device=dev) for _ in range(3))
        lib.pointwise_block_forward(B, c, ch, S, act, p(conv), p(x), p(ws), p(bs), p(w1), p(b1), p(w2), p(b2), p(gt), p(y), p(pre),
                                    p(out), st)
        gx, acc = torch.empty_like(x), torch.empty_like(x)
        gw1, gw2, gb1, gb2, ggt = (torch.empty_like(t) for t in (w1, w2, b1, b2, gt))
        wsb = torch.empty(lib.pointwise_mlp_workspace_bytes(B, c, ch, c, S, 1), dtype=torch.uint8, device=dev)
        lib.pointwise_mlp_backward(B, c, ch, c, S, act, p(y), p(w1), p(b1), p(w2), p(b2), p(x), p(gt), p(gout), p(gx), p(gw1), p(gb1),
                                   p(gw2), p(gb2), p(acc), p(ggt), p(wsb), st, x_pre=p(pre))
        torch.cuda.synchronize()
        res[act] = (y, pre, out, gx, acc, gw1, gw2, gb1, gb2, ggt)
    a, d = res[_lib.SC_ACT_GELU], res[_lib.SC_ACT_GELU_DGRAD]
    # (the two forms evaluate the same expressions in different kernels: hipcc may contract them differently -- last-bit
    #  differences on the device, identical bits in host emulation, tests/test_emu_pmlp.py)
    close = lambda u, v: rel_l2(u.cpu().numpy(), v.cpu().numpy()) < 1e-6
    assert close(a[0], d[0]) and close(a[2], d[2])
    t = a[1].double().requires_grad_(True)
    F.gelu(t).backward(torch.ones_like(t))
    assert rel_l2(d[1].cpu().numpy(), t.grad.cpu().numpy()) < TOL
    for i in range(3, 10):
        assert close(a[i], d[i]), i


@pytest.mark.parametrize("act", [_lib.SC_ACT_GELU_DGRAD, _lib.SC_ACT_NONE])
def test_block_backward_with_the_linear_skip_on_device(lib, act):
    """sc_pointwise_block_backward (round 6) at the metric block's channel counts against the two passes it replaces and,
    for the gradient of the block input, against float64 torch."""
    dev = torch.device("cuda:0")
    c, ch, B, S = 64, 32, 4, 4096
    g = torch.Generator(device="cpu").manual_seed(17 + act)
    mk = lambda *sh: torch.randn(*sh, generator=g).to(dev)
    x, conv, gout = mk(B, c, S), mk(B, c, S), mk(B, c, S)
    ws, w1, w2 = mk(c, c) / c ** 0.5, mk(ch, c) / c ** 0.5, mk(c, ch) / ch ** 0.5
    bs, b1, b2, gt = mk(c), mk(ch), mk(c), mk(c)
    st = torch.cuda.current_stream().cuda_stream
    p = lambda t: 0 if t is None else t.data_ptr()
    y, out = torch.empty_like(x), torch.empty_like(x)
    pre = torch.empty_like(x) if act else None
    lib.pointwise_block_forward(B, c, ch, S, act, p(conv), p(x), p(ws), p(bs), p(w1), p(b1), p(w2), p(b2), p(gt), p(y), p(pre), p(out), st)
    new = lambda t: torch.full_like(t, float("nan"))
    wsb = torch.empty(lib.pointwise_mlp_workspace_bytes(B, c, ch, c, S, 1), dtype=torch.uint8, device=dev)
    gz0, acc, gin0 = new(x), new(x), new(x)
    a = [new(t) for t in (w1, b1, w2, b2, gt)]
    lib.pointwise_mlp_backward(B, c, ch, c, S, act, p(y), p(w1), p(b1), p(w2), p(b2), p(x), p(gt), p(gout), p(gz0), p(a[0]), p(a[1]),
                               p(a[2]), p(a[3]), p(acc), p(a[4]), p(wsb), st, x_pre=p(pre))
    glw0, glb0 = new(ws), new(bs)
    wl = torch.empty(lib.pointwise_linear_workspace_bytes(B, c, c, S), dtype=torch.uint8, device=dev)
    lib.pointwise_linear_backward(B, c, c, S, p(x), p(ws), p(gz0), p(gin0), p(glw0), p(glb0), p(wl), st, addend=p(acc))
    gz1, gin1 = new(x), new(x)
    d = [new(t) for t in (w1, b1, w2, b2, gt)]
    lib.pointwise_block_backward(B, c, ch, S, act, p(y), p(pre), p(x), p(ws), p(w1), p(b1), p(w2), p(b2), p(gt), p(gout), p(gz1), p(gin1),
                                 p(d[0]), p(d[1]), p(d[2]), p(d[3]), p(d[4]), p(wsb), st)
    glw1, glb1 = new(ws), new(bs)
    wx = torch.empty(lib.pointwise_linear_workspace_bytes_ex(B, c, c, S), dtype=torch.uint8, device=dev)
    lib.pointwise_linear_backward_ex(B, c, c, S, 0, p(x), p(ws), p(gz1), 0, 0, 0, 0, 0, 0, p(glw1), p(glb1), 0, 0, p(wx), st)
    torch.cuda.synchronize()
    close = lambda u, v: rel_l2(u.cpu().numpy(), v.cpu().numpy()) < 1e-6
    assert close(gz1, gz0) and close(gin1, gin0) and close(glw1, glw0) and close(glb1, glb0)
    for u, v in zip(d, a):
        assert close(u, v)
    # the gradient of the block input outside the spectral convolution, float64: d/dx of <out, gout> with conv held fixed
    xd = x.double().requires_grad_(True)
    sd = conv.double() + torch.einsum("oc,bcs->bos", ws.double(), xd) + bs.double()[None, :, None]
    yd = F.gelu(sd) if act else sd
    hd = F.gelu(torch.einsum("hc,bcs->bhs", w1.double(), yd) + b1.double()[None, :, None])
    zd = torch.einsum("oh,bhs->bos", w2.double(), hd) + b2.double()[None, :, None] + gt.double()[None, :, None] * xd
    (F.gelu(zd) if act else zd).backward(gout.double())
    assert rel_l2(gin1.cpu().numpy(), xd.grad.cpu().numpy()) < TOL
