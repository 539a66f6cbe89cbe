"""CPU tier: the pointwise half of an FNO block in one pass (sc_kernels_pmlp.h) in host
emulation against the same computation in torch: out = act(W2 gelu(W1 x + b1) + b2 + gate * skip) -- the reference's
ChannelMLP (channel_mlp.py:82-119) + soft-gating skip (skip_connections.py:53-130) + closing non-linearity
(fno_block.py:399-412)."""
import pytest
import torch
import torch.nn.functional as F

from engine_runner import emu_lib, rel_l2
from neuraloperator_amd import _lib

TOL = 2e-6


@pytest.fixture(scope="module")
def lib():
    return emu_lib()


def _ref(x, w1, b1, w2, b2, skip, gate, act):
    h = F.gelu(torch.einsum("hc,bcs->bhs", w1.double(), x.double()) + (0 if b1 is None else b1.double()[None, :, None]))
    z = torch.einsum("oh,bhs->bos", w2.double(), h) + (0 if b2 is None else b2.double()[None, :, None])
    if skip is not None:
        z = z + gate.double()[None, :, None] * skip.double()
    return F.gelu(z) if act else z


@pytest.mark.parametrize("chans", [(32, 32, 32), (64, 32, 64), (64, 64, 64), (128, 64, 128)], ids=str)
@pytest.mark.parametrize("gate,act,bias", [(True, 1, True), (False, 0, True), (True, 0, False)])
def test_pointwise_mlp_forward(lib, chans, gate, act, bias):
    ci, ch, co = chans
    g = torch.Generator().manual_seed(ci + ch + act)
    B, S = 3, 96                                        # 9 tiles of 32 pixels: more tiles than one wave round
    x = torch.randn(B, ci, S, generator=g)
    w1 = torch.randn(ch, ci, generator=g) / ci ** 0.5
    w2 = torch.randn(co, ch, generator=g) / ch ** 0.5
    b1 = torch.randn(ch, generator=g) if bias else None
    b2 = torch.randn(co, generator=g) if bias else None
    skip = torch.randn(B, co, S, generator=g) if gate else None
    gt = torch.randn(co, generator=g) if gate else None
    out = torch.full((B, co, S), float("nan"))
    p = lambda t: 0 if t is None else t.data_ptr()
    lib.pointwise_mlp_forward(B, ci, ch, co, S, act, p(x), p(w1), p(b1), p(w2), p(b2), p(skip), p(gt), p(out), 0)
    assert rel_l2(out.numpy(), _ref(x, w1, b1, w2, b2, skip, gt, act).numpy()) < TOL


def test_backward_shapes(lib):
    assert lib.pointwise_mlp_workspace_bytes(2, 64, 32, 64, 96, 1) > 0
    assert lib.pointwise_mlp_workspace_bytes(2, 128, 64, 128, 96, 1) == 0     # forward kernel only at 128 channels


def test_pointwise_mlp_argument_checks(lib):
    x = torch.zeros(1, 64, 32)
    w1, w2, out = torch.zeros(32, 64), torch.zeros(64, 32), torch.zeros(1, 64, 32)
    with pytest.raises(_lib.EngineError):                # 48 channels: no kernel
        lib.pointwise_mlp_forward(1, 48, 32, 48, 32, 0, x.data_ptr(), w1.data_ptr(), 0, w2.data_ptr(), 0, 0, 0, out.data_ptr(), 0)
    with pytest.raises(_lib.EngineError):                # spatial not a multiple of 32
        lib.pointwise_mlp_forward(1, 64, 32, 64, 40, 0, x.data_ptr(), w1.data_ptr(), 0, w2.data_ptr(), 0, 0, 0, out.data_ptr(), 0)
    with pytest.raises(_lib.EngineError):                # a gate without its source
        lib.pointwise_mlp_forward(1, 64, 32, 64, 32, 0, x.data_ptr(), w1.data_ptr(), 0, w2.data_ptr(), 0, 0, w1.data_ptr(), out.data_ptr(), 0)


BWD_CASES = [((64, 32, 64), True, 1, True), ((64, 32, 64), False, 0, True), ((64, 32, 64), True, 0, False),
             ((64, 32, 64), False, 1, False), ((32, 32, 32), True, 1, True), ((64, 64, 64), True, 1, False),
             ]


@pytest.mark.parametrize("chans,gate,act,bias", BWD_CASES, ids=[f"{c[0]}-{c[1]}-{c[2]}_g{int(g)}a{a}b{int(b)}" for c, g, a, b in BWD_CASES])
def test_pointwise_mlp_backward(lib, chans, gate, act, bias):
    """Every gradient of the fused pass (recomputation inside the tile, pixel contraction of the weight gradients
    through the LDS transposes, fixed-order reduction of the per-workgroup partials) against torch autograd of the
    float64 composition."""
    ci, ch, co = chans
    g = torch.Generator().manual_seed(7 * ci + ch + 3 * act)
    B, S = 2, 96                                        # 6 tiles: 2 workgroups, the last with two idle waves
    mk = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc)
    x, gout = mk(B, ci, S), mk(B, co, S)
    w1, w2 = mk(ch, ci, sc=ci ** -0.5), mk(co, ch, sc=ch ** -0.5)
    b1, b2 = (mk(ch), mk(co)) if bias else (None, None)
    skip, gt = (mk(B, co, S), mk(co)) if gate else (None, None)
    leaves = [t.double().requires_grad_(True) if t is not None else None for t in (x, w1, b1, w2, b2, skip, gt)]
    xd, w1d, b1d, w2d, b2d, skd, gtd = leaves
    h = F.gelu(torch.einsum("hc,bcs->bhs", w1d, xd) + (0 if b1d is None else b1d[None, :, None]))
    z = torch.einsum("oh,bhs->bos", w2d, h) + (0 if b2d is None else b2d[None, :, None])
    if gate:
        z = z + gtd[None, :, None] * skd
    out = F.gelu(z) if act else z
    out.backward(gout.double())
    gx, gw1, gw2 = torch.full_like(x, float("nan")), torch.full_like(w1, float("nan")), torch.full_like(w2, float("nan"))
    gb1 = torch.full((ch,), float("nan")) if bias else None
    gb2 = torch.full((co,), float("nan")) if bias else None
    gsk = torch.full((B, co, S), float("nan")) if gate else None
    ggt = torch.full((co,), float("nan")) if gate else None
    ws = torch.empty(lib.pointwise_mlp_workspace_bytes(B, ci, ch, co, S, act), dtype=torch.uint8)
    p = lambda t: 0 if t is None else t.data_ptr()
    lib.pointwise_mlp_backward(B, ci, ch, co, S, act, p(x), p(w1), p(b1), p(w2), p(b2), p(skip), p(gt), p(gout),
                               p(gx), p(gw1), p(gb1), p(gw2), p(gb2), p(gsk), p(ggt), p(ws), 0)
    tol = 1e-5
    assert rel_l2(gx.numpy(), xd.grad.numpy()) < tol
    assert rel_l2(gw1.numpy(), w1d.grad.numpy()) < tol
    assert rel_l2(gw2.numpy(), w2d.grad.numpy()) < tol
    if bias:
        assert rel_l2(gb1.numpy(), b1d.grad.numpy()) < tol and rel_l2(gb2.numpy(), b2d.grad.numpy()) < tol
    if gate:
        assert rel_l2(gsk.numpy(), skd.grad.numpy()) < tol and rel_l2(ggt.numpy(), gtd.grad.numpy()) < tol


@pytest.mark.parametrize("c", [32, 64, 128])
@pytest.mark.parametrize("bias", [True, False])
def test_pointwise_linear(lib, c, bias):
    """The block's 1 x 1 linear skip in one pass each way against torch (float64)."""
    g = torch.Generator().manual_seed(c + bias)
    B, S = 2, 96
    x, go = torch.randn(B, c, S, generator=g), torch.randn(B, c, S, generator=g)
    w = torch.randn(c, c, generator=g) / c ** 0.5
    b = torch.randn(c, generator=g) if bias else None
    p = lambda t: 0 if t is None else t.data_ptr()
    out = torch.full((B, c, S), float("nan"))
    lib.pointwise_linear_forward(B, c, c, S, p(x), p(w), p(b), p(out), 0)
    xd, wd = x.double().requires_grad_(True), w.double().requires_grad_(True)
    bd = None if b is None else b.double().requires_grad_(True)
    ref = torch.einsum("oc,bcs->bos", wd, xd) + (0 if bd is None else bd[None, :, None])
    assert rel_l2(out.numpy(), ref.detach().numpy()) < TOL
    if c == 128:
        assert lib.pointwise_linear_workspace_bytes(B, c, c, S) == 0        # forward only at 128 channels
        return
    ref.backward(go.double())
    gx, gw = torch.full_like(x, float("nan")), torch.full_like(w, float("nan"))
    gb = torch.full((c,), float("nan")) if bias else None
    ws = torch.empty(lib.pointwise_linear_workspace_bytes(B, c, c, S), dtype=torch.uint8)
    lib.pointwise_linear_backward(B, c, c, S, p(x), p(w), p(go), p(gx), p(gw), p(gb), p(ws), 0)
    add = torch.randn(B, c, S, generator=g)                                    # gx = W^T g + addend in the store path
    gx2 = torch.full_like(x, float("nan"))
    lib.pointwise_linear_backward(B, c, c, S, p(x), p(w), p(go), p(gx2), p(gw), p(gb), p(ws), 0, addend=p(add))
    assert rel_l2(gx2.numpy(), (xd.grad + add.double()).numpy()) < 1e-5
    assert rel_l2(gx.numpy(), xd.grad.numpy()) < 1e-5 and rel_l2(gw.numpy(), wd.grad.numpy()) < 1e-5
    if bias:
        assert rel_l2(gb.numpy(), bd.grad.numpy()) < 1e-5


@pytest.mark.parametrize("chans", [(32, 32), (64, 32), (64, 64)], ids=str)
@pytest.mark.parametrize("act,bias", [(1, True), (0, True), (1, False)])
def test_pointwise_block_forward(lib, chans, act, bias):
    """The whole pointwise side of a default block in one pass (session 2, k_pblock_fwd):
    s = conv + (Ws x + bs), y = act(s), out = act(W2 gelu(W1 y + b1) + b2 + gate x) -- fno_block.py:392-412 with the
    linear skip (skip_connections.py:119-169), ChannelMLP and soft-gating skip; y and s are outputs too.  Also against
    the three passes it replaces (same operand order in the skip product: y and s agree to the bit)."""
    c, ch = chans
    g = torch.Generator().manual_seed(c + ch + act)
    B, S = 3, 96
    x, conv = torch.randn(B, c, S, generator=g), torch.randn(B, c, S, generator=g)
    ws = torch.randn(c, c, generator=g) / c ** 0.5
    w1 = torch.randn(ch, c, generator=g) / c ** 0.5
    w2 = torch.randn(c, ch, generator=g) / ch ** 0.5
    bs = torch.randn(c, generator=g) if bias else None
    b1 = torch.randn(ch, generator=g) if bias else None
    b2 = torch.randn(c, generator=g) if bias else None
    gt = torch.randn(c, generator=g)
    y, pre, out = (torch.full((B, c, S), float("nan")) for _ in range(3))
    p = lambda t: 0 if t is None else t.data_ptr()
    lib.pointwise_block_forward(B, c, ch, S, act, p(conv), p(x), p(ws), p(bs), p(w1), p(b1), p(w2), p(b2), p(gt), p(y),
                                p(pre) if act else 0, p(out), 0)
    s_ref = conv.double() + torch.einsum("oc,bcs->bos", ws.double(), x.double()) + (0 if bs is None else bs.double()[None, :, None])
    y_ref = F.gelu(s_ref) if act else s_ref
    out_ref = _ref(y_ref.float(), w1, b1, w2, b2, x, gt, act)
    assert rel_l2(y.numpy(), y_ref.numpy()) < TOL
    if act:
        assert rel_l2(pre.numpy(), s_ref.numpy()) < TOL
    assert rel_l2(out.numpy(), out_ref.numpy()) < 2 * TOL
    # the passes it replaces
    skip = torch.empty_like(x)
    lib.pointwise_linear_forward(B, c, c, S, p(x), p(ws), p(bs), p(skip), 0)
    s3 = conv + skip
    if act:
        assert torch.equal(pre, s3)
    else:
        assert torch.equal(y, s3)


def test_pointwise_block_argument_checks(lib):
    z = torch.zeros(1, 64, 32)
    w = torch.zeros(64, 64)
    with pytest.raises(_lib.EngineError):                # GELU without the pre-activation buffer
        lib.pointwise_block_forward(1, 64, 64, 32, 1, z.data_ptr(), z.data_ptr(), w.data_ptr(), 0, w.data_ptr(), 0, w.data_ptr(), 0,
                                    w.data_ptr(), z.data_ptr(), 0, z.data_ptr(), 0)
    with pytest.raises(_lib.EngineError):                # 128 channels: no kernel
        lib.pointwise_block_forward(1, 128, 64, 32, 0, z.data_ptr(), z.data_ptr(), w.data_ptr(), 0, w.data_ptr(), 0, w.data_ptr(), 0,
                                    w.data_ptr(), z.data_ptr(), 0, z.data_ptr(), 0)


@pytest.mark.parametrize("chans", [(64, 32), (32, 32)], ids=str)
def test_block_pass_stores_the_derivative_and_the_backward_multiplies_by_it(lib, chans):
    """SC_ACT_GELU_DGRAD (round 6): sc_pointwise_block_forward writes gelu'(s) where SC_ACT_GELU writes s -- y and out are the
    same bits -- and sc_pointwise_mlp_backward_ex with that buffer as x_pre returns what it returns for the pre-activation
    itself (the derivative from the shared evaluation and the one sc_gelu_grad computes are the same expression)."""
    c, ch = chans
    g = torch.Generator().manual_seed(c + ch)
    B, S = 2, 96
    mk = lambda *sh: torch.randn(*sh, generator=g)
    x, conv, gout = mk(B, c, S), mk(B, c, S), mk(B, c, S)
    ws, w1, w2 = mk(c, c) / c ** 0.5, mk(ch, c) / c ** 0.5, mk(c, ch) / ch ** 0.5
    bs, b1, b2, gt = mk(c), mk(ch), mk(c), mk(c)
    p = lambda t: 0 if t is None else t.data_ptr()
    res = {}
    for act in (_lib.SC_ACT_GELU, _lib.SC_ACT_GELU_DGRAD):
        y, pre, out = (torch.full((B, c, S), float("nan")) for _ in range(3))
        lib.pointwise_block_forward(B, c, ch, S, act, p(conv), p(x), p(ws), p(bs), p(w1), p(b1), p(w2), p(b2), p(gt), p(y),
                                    p(pre), p(out), 0)
        gx, acc = torch.empty_like(x), torch.empty_like(x)
        gw1, gw2, gb1, gb2, ggt = torch.empty_like(w1), torch.empty_like(w2), torch.empty_like(b1), torch.empty_like(b2), torch.empty_like(gt)
        wsb = torch.empty(lib.pointwise_mlp_workspace_bytes(B, c, ch, c, S, 1), dtype=torch.uint8)
        lib.pointwise_mlp_backward(B, c, ch, c, S, act, p(y), p(w1), p(b1), p(w2), p(b2), p(x), p(gt), p(gout), p(gx), p(gw1), p(gb1),
                                   p(gw2), p(gb2), p(acc), p(ggt), p(wsb), 0, x_pre=p(pre))
        res[act] = (y, pre, out, gx, acc, gw1, gw2, gb1, gb2, ggt)
    a, d = res[_lib.SC_ACT_GELU], res[_lib.SC_ACT_GELU_DGRAD]
    assert torch.equal(a[0], d[0]) and torch.equal(a[2], d[2])                       # y, out
    t = a[1].double().requires_grad_(True)                                           # s -> gelu'(s)
    F.gelu(t).backward(torch.ones_like(t))
    assert rel_l2(d[1].numpy(), t.grad.numpy()) < TOL
    for i in range(3, 10):
        assert torch.equal(a[i], d[i]), i
    with pytest.raises(_lib.EngineError):                # the derivative buffer is required
        lib.pointwise_mlp_backward(B, c, ch, c, S, _lib.SC_ACT_GELU_DGRAD, p(a[0]), p(w1), p(b1), p(w2), p(b2), p(x), p(gt), p(gout),
                                   p(gx), p(gw1), p(gb1), p(gw2), p(gb2), p(acc), p(ggt), p(wsb), 0)


@pytest.mark.parametrize("chans", [(64, 32), (32, 32)], ids=str)
@pytest.mark.parametrize("act", [_lib.SC_ACT_GELU_DGRAD, _lib.SC_ACT_GELU, _lib.SC_ACT_NONE])
def test_block_backward_with_the_linear_skip_riding_along(lib, chans, act):
    """sc_pointwise_block_backward (round 6: k_pmlp_bwd<.., LIN>) against the two passes it replaces -- the MLP pass
    (sc_pointwise_mlp_backward_ex) and the linear skip's backward with the soft-gating gradient as its addend
    (sc_pointwise_linear_backward): gz and the MLP's parameter gradients to the bit, gin = W_s^T gz + gate (.) g_z up to the
    order of its additions; the skip's weight gradient from sc_pointwise_linear_backward_ex(gx = NULL) to the bit."""
    c, ch = chans
    g = torch.Generator().manual_seed(c + ch + act)
    B, S = 2, 96
    mk = lambda *sh: torch.randn(*sh, generator=g)
    x, conv, gout = mk(B, c, S), mk(B, c, S), mk(B, c, S)
    ws, w1, w2 = mk(c, c) / c ** 0.5, mk(ch, c) / c ** 0.5, mk(c, ch) / ch ** 0.5
    bs, b1, b2, gt = mk(c), mk(ch), mk(c), mk(c)
    p = lambda t: 0 if t is None else t.data_ptr()
    y, out = torch.empty(B, c, S), torch.empty(B, c, S)
    pre = torch.empty(B, c, S) if act else None
    lib.pointwise_block_forward(B, c, ch, S, act, p(conv), p(x), p(ws), p(bs), p(w1), p(b1), p(w2), p(b2), p(gt), p(y), p(pre), p(out), 0)
    new = lambda t: torch.full_like(t, float("nan"))
    wsb = torch.empty(lib.pointwise_mlp_workspace_bytes(B, c, ch, c, S, 1), dtype=torch.uint8)
    # the two passes
    gz0, acc, gin0 = new(x), new(x), new(x)
    a = [new(t) for t in (w1, b1, w2, b2, gt)]
    lib.pointwise_mlp_backward(B, c, ch, c, S, act, p(y), p(w1), p(b1), p(w2), p(b2), p(x), p(gt), p(gout), p(gz0), p(a[0]), p(a[1]),
                               p(a[2]), p(a[3]), p(acc), p(a[4]), p(wsb), 0, x_pre=p(pre))
    glw0, glb0 = new(ws), new(bs)
    wl = torch.empty(lib.pointwise_linear_workspace_bytes(B, c, c, S), dtype=torch.uint8)
    lib.pointwise_linear_backward(B, c, c, S, p(x), p(ws), p(gz0), p(gin0), p(glw0), p(glb0), p(wl), 0, addend=p(acc))
    # the fused pass + the skip's weight gradient
    assert lib.pointwise_block_backward_supported(B, c, ch, S)
    gz1, gin1 = new(x), new(x)
    d = [new(t) for t in (w1, b1, w2, b2, gt)]
    lib.pointwise_block_backward(B, c, ch, S, act, p(y), p(pre), p(x), p(ws), p(w1), p(b1), p(w2), p(b2), p(gt), p(gout), p(gz1), p(gin1),
                                 p(d[0]), p(d[1]), p(d[2]), p(d[3]), p(d[4]), p(wsb), 0)
    glw1, glb1 = new(ws), new(bs)
    wx = torch.empty(lib.pointwise_linear_workspace_bytes_ex(B, c, c, S), dtype=torch.uint8)
    lib.pointwise_linear_backward_ex(B, c, c, S, 0, p(x), p(ws), p(gz1), 0, 0, 0, 0, 0, 0, p(glw1), p(glb1), 0, 0, p(wx), 0)
    assert torch.equal(gz1, gz0)
    for u, v in zip(d, a):
        assert torch.equal(u, v)
    assert rel_l2(gin1.numpy(), gin0.numpy()) < 1e-6
    assert rel_l2(glw1.numpy(), glw0.numpy()) < 1e-6 and rel_l2(glb1.numpy(), glb0.numpy()) < 1e-6
    assert not lib.pointwise_block_backward_supported(B, 64, 64, S)                    # tables + scratch + gradients > 160 KB
