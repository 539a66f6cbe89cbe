"""CPU tier: the block epilogue -- y = act(spectral_conv(x) + skip), addition and activation
in the store path of the inverse transform (fused 2-D kernels) or as one streaming pass behind it (other shapes) --
in host emulation against the oracle composition gelu(forward_torch(x) + skip), and the drop-in module's
``forward_fused`` with autograd (gradients of x, weight, bias and skip) against torch autograd of that composition."""
import numpy as np
import pytest
import torch

from emu_engine import engine_on_emulation
from engine_runner import emu_lib, rel_l2
from neuraloperator_amd import _lib
from neuraloperator_amd.modes import kept_block
from oracle import spectral_oracle as so

TOL = 1e-5


@pytest.fixture(scope="module")
def lib():
    return emu_lib()


@pytest.mark.parametrize("spatial,modes,act", [((64, 256), (8, 8), "gelu"),      # fused 2-D kernels: in the store path
                                               ((64, 256), (8, 8), None),
                                               ((12, 10), (6, 6), "gelu"),       # size-agnostic passes + epilogue pass
                                               ((6, 8, 10), (4, 4, 4), "gelu")])
def test_layer_forward_ex(lib, spatial, modes, act):
    torch.manual_seed(5)
    b, ci, co = 2, 2, 3
    nm = so.halve_last(modes)
    x = torch.randn(b, ci, *spatial)
    w = torch.randn(ci, co, *nm, dtype=torch.cfloat) * 0.5
    bias = torch.randn(co, *(1,) * len(spatial))
    skip = torch.randn(b, co, *spatial)
    kept, w_start = kept_block(list(spatial), nm, nm)
    plan = lib.plan_create(list(spatial), kept)
    assert lib.plan_is_fast(plan) == (spatial == (64, 256))
    L = lib.layer_desc(b, ci, co, list(w.shape[2:]), w_start)
    ws = torch.empty(lib.layer_workspace_bytes(plan, L), dtype=torch.uint8)
    y = torch.full((b, co, *spatial), float("nan"))
    pre = torch.full_like(y, float("nan"))
    xhat = torch.empty(b, ci, *kept, 2)
    wv = torch.view_as_real(w.contiguous())
    lib.layer_forward_ex(plan, L, x.data_ptr(), wv.data_ptr(), bias.reshape(-1).contiguous().data_ptr(),
                         skip.data_ptr(), pre.data_ptr() if act else 0,
                         _lib.SC_ACT_GELU if act else _lib.SC_ACT_NONE, y.data_ptr(), xhat.data_ptr(), ws.data_ptr())
    z = so.forward_torch(x, w, bias, nm, nm) + skip
    want = torch.nn.functional.gelu(z) if act else z
    assert rel_l2(y.numpy(), want.numpy()) < TOL
    if act:
        assert rel_l2(pre.numpy(), z.numpy()) < TOL
    # a NULL skip is the plain layer
    y0 = torch.empty_like(y)
    lib.layer_forward_ex(plan, L, x.data_ptr(), wv.data_ptr(), bias.reshape(-1).contiguous().data_ptr(), 0, 0,
                         _lib.SC_ACT_GELU, y0.data_ptr(), xhat.data_ptr(), ws.data_ptr())
    assert rel_l2(y0.numpy(), so.forward_torch(x, w, bias, nm, nm).numpy()) < TOL
    lib.plan_destroy(plan)


@pytest.mark.parametrize("spatial", [(64, 256), (12, 10)])
def test_gelu_of_the_epilogue_against_torch(lib, spatial):
    """The activation itself, pinned for |v| <= 6 on BOTH routes (fused store path / stand-alone pass):
    zero weight and bias make the layer's output gelu(skip).  Bars (include/sc_engine.h, SC_ACT_GELU): absolute error
    <= 0.5 |v| 7e-7 (A&S 7.1.26's 1.5e-7 + fp32 round-off of its evaluation), and -- because the negative tail is
    taken from erfc directly -- a RELATIVE error below 1 % down to v = -6, where 1 + erf(v / sqrt 2) = 2e-9 (the
    1 + erf form had lost every digit there); the two routes agree bit for bit;
    preact together with SC_ACT_NONE is refused."""
    b, c = 1, 2
    nm = so.halve_last((4, 4))
    n = b * c * spatial[0] * spatial[1]
    v = torch.linspace(-6.0, 6.0, n).reshape(b, c, *spatial).contiguous()
    x = torch.zeros(b, c, *spatial)
    w = torch.zeros(c, c, *nm, dtype=torch.cfloat)
    kept, w_start = kept_block(list(spatial), nm, nm)
    plan = lib.plan_create(list(spatial), kept)
    L = lib.layer_desc(b, c, c, list(w.shape[2:]), w_start)
    ws = torch.empty(lib.layer_workspace_bytes(plan, L), dtype=torch.uint8)
    y = torch.full((b, c, *spatial), float("nan"))
    xhat = torch.empty(b, c, *kept, 2)
    wv = torch.view_as_real(w.contiguous())
    zb = torch.zeros(c)
    lib.layer_forward_ex(plan, L, x.data_ptr(), wv.data_ptr(), zb.data_ptr(), v.data_ptr(), 0, _lib.SC_ACT_GELU,
                         y.data_ptr(), xhat.data_ptr(), ws.data_ptr())
    want = torch.nn.functional.gelu(v.double())
    err = (y.double() - want).abs()
    assert bool((err <= 0.5 * v.abs().double() * 7e-7 + 1e-12).all()), float(err.max())
    rel = err / want.abs().clamp_min(1e-300)
    assert float(rel[v < -0.1].max()) < 1e-2, float(rel[v < -0.1].max())     # erfc form: no cancellation in the tail
    assert float(rel[v > 0.1].max()) < 1e-6
    with pytest.raises(RuntimeError):                                            # preact + SC_ACT_NONE
        lib.layer_forward_ex(plan, L, x.data_ptr(), wv.data_ptr(), zb.data_ptr(), v.data_ptr(), y.data_ptr(),
                             _lib.SC_ACT_NONE, y.data_ptr(), xhat.data_ptr(), ws.data_ptr())
    lib.plan_destroy(plan)
    # the other route on the same values: same bits
    other = (12, 10) if spatial == (64, 256) else (64, 256)
    m = min(n, b * c * other[0] * other[1])
    vo = torch.zeros(b * c * other[0] * other[1])
    vo[:m] = v.reshape(-1)[:m]
    vo = vo.reshape(b, c, *other).contiguous()
    kept2, w_start2 = kept_block(list(other), nm, nm)
    plan2 = lib.plan_create(list(other), kept2)
    L2 = lib.layer_desc(b, c, c, list(w.shape[2:]), w_start2)
    ws2 = torch.empty(lib.layer_workspace_bytes(plan2, L2), dtype=torch.uint8)
    y2 = torch.empty(b, c, *other)
    xhat2 = torch.empty(b, c, *kept2, 2)
    x0 = torch.zeros(b, c, *other)              # kept alive across the call (a temporary's storage is freed before it runs)
    lib.layer_forward_ex(plan2, L2, x0.data_ptr(), wv.data_ptr(), zb.data_ptr(), vo.data_ptr(), 0,
                         _lib.SC_ACT_GELU, y2.data_ptr(), xhat2.data_ptr(), ws2.data_ptr())
    assert torch.equal(y2.reshape(-1)[:m], y.reshape(-1)[:m])
    lib.plan_destroy(plan2)


@pytest.mark.parametrize("spatial", [(64, 256), (12, 10)])
def test_module_forward_fused_with_autograd(spatial):
    from neuraloperator_amd import SpectralConv
    torch.manual_seed(6)
    conv = SpectralConv(2, 3, (8, 8))
    x = torch.randn(2, 2, *spatial, requires_grad=True)
    skip = torch.randn(2, 3, *spatial, requires_grad=True)
    g = torch.randn(2, 3, *spatial)
    with engine_on_emulation():
        out = conv.forward_fused(x, skip, "gelu")
        out.backward(g)
    got = [out.detach(), x.grad.clone(), skip.grad.clone(), conv.weight.tensor.grad.clone(), conv.bias.grad.clone()]
    xr, sr = x.detach().clone().requires_grad_(True), skip.detach().clone().requires_grad_(True)
    wr = conv.weight.tensor.detach().clone().requires_grad_(True)
    br = conv.bias.detach().clone().requires_grad_(True)
    ref = torch.nn.functional.gelu(so.forward_torch(xr, wr, br, conv.n_modes, conv.max_n_modes) + sr)
    ref.backward(g)
    want = [ref.detach(), xr.grad, sr.grad, wr.grad, br.grad]
    for a, b_, name in zip(got, want, ("out", "gx", "gskip", "gw", "gbias")):
        assert rel_l2(a.numpy(), b_.numpy()) < TOL, name
