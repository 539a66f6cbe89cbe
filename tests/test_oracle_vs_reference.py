"""Pin the oracle against the VERBATIM reference module (only where /root/reference
exists, i.e. the build container; skipped on the GPU box)."""
import pytest
import torch

from oracle import ref_verbatim
from oracle import spectral_oracle as so

pytestmark = pytest.mark.skipif(not ref_verbatim.available(),
                                reason="/root/reference not present")

CASES = [((12,), (6,), None, None), ((12, 10), (6, 4), None, None), ((9, 11), (5, 7), None, None),
         ((8, 6, 10), (4, 4, 6), None, None), ((16, 16), (6, 6), (8, 8), None),
         ((16, 16), (8, 8), None, (6, 4)), ((6, 6), (8, 8), None, None),
         ((12, 10), (7, 5), None, (4, 3)), ((6, 6, 6, 6), (4, 4, 4, 4), None, None)]


@pytest.mark.parametrize("spatial,nm,maxm,newm", CASES)
def test_restatements_match_verbatim(spatial, nm, maxm, newm):
    ref = ref_verbatim.load_reference()
    torch.manual_seed(0)
    conv = ref.SpectralConv(3, 4, nm, max_n_modes=maxm)
    if newm is not None:
        conv.n_modes = newm
    x = torch.randn(2, 3, *spatial, requires_grad=True)
    y = conv(x)
    g = torch.randn_like(y)
    y.backward(g)
    w = conv.weight.to_tensor().detach()
    b = conv.bias.detach()
    x2 = x.detach().clone().requires_grad_(True)
    w2 = w.clone().requires_grad_(True)
    b2 = b.clone().requires_grad_(True)
    y2 = so.forward_torch(x2, w2, b2, conv.n_modes, conv.max_n_modes)
    y2.backward(g)
    assert so.rel_l2(y2.detach().numpy(), y.detach().numpy()) < 1e-7
    assert so.rel_l2(x2.grad.numpy(), x.grad.numpy()) < 1e-7
    assert so.rel_l2(w2.grad.numpy(), conv.weight.tensor.grad.numpy()) < 1e-7
    y3, _ = so.forward_np64(x.detach().numpy(), w.numpy(), b.numpy(), conv.n_modes, conv.max_n_modes)
    gx3, gw3, gb3 = so.backward_np64(x.detach().numpy(), w.numpy(), g.numpy(),
                                     conv.n_modes, conv.max_n_modes)
    assert so.rel_l2(y3, y.detach().numpy()) < 1e-6
    assert so.rel_l2(gx3, x.grad.numpy()) < 1e-6
    assert so.rel_l2(gw3, conv.weight.tensor.grad.numpy()) < 1e-6
    assert so.rel_l2(gb3, conv.bias.grad.numpy()) < 1e-6


@pytest.mark.parametrize("fac", ["Tucker", "CP"])
@pytest.mark.parametrize("dim", [1, 2, 3])
def test_factorized_contract_matches_verbatim(fac, dim):
    """the reference's own self-consistency identity,
    neuralop/layers/tests/test_spectral_convolution.py:54-65, through our contraction order."""
    ref = ref_verbatim.load_reference()
    torch.manual_seed(1)
    modes = (10, 8, 6)[:dim]
    conv = ref.SpectralConv(3, 3, modes, bias=False, factorization=fac, implementation="factorized")
    x = torch.randn(2, 3, *(12,) * dim)
    y = conv(x)
    w = conv.weight
    facs = [f.detach() for f in w.factors]
    if fac == "Tucker":
        contract = lambda xk, wk: so.contract_tucker(xk, w.core.detach(), facs)
        dense = so.reconstruct_tucker(w.core.detach(), facs)
    else:
        contract = lambda xk, wk: so.contract_cp(xk, w.weights.detach(), facs)
        dense = so.reconstruct_cp(w.weights.detach(), facs)
    y2 = so.forward_torch(x, dense, None, conv.n_modes, conv.max_n_modes, contract=contract)
    assert so.rel_l2(y2.numpy(), y.detach().numpy()) < 2e-6
    assert so.rel_l2(dense.numpy(), w.to_tensor().detach().numpy()) < 1e-6
