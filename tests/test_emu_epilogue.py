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
