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


VARIANTS = [dict(separable=True), dict(factorization="TT", rank=0.5, implementation="factorized"),
            dict(resolution_scaling_factor=2), dict(resolution_scaling_factor=[0.5, 1.5]), dict(complex_data=True),
            dict(complex_data=True, resolution_scaling_factor=0.5), dict(separable=True, factorization="Tucker", rank=0.5)]


@pytest.mark.parametrize("kw", VARIANTS, ids=lambda k: "-".join(f"{a}={b}" for a, b in k.items()))
@pytest.mark.parametrize("spatial,nm", [((16, 12), (8, 6)), ((9, 11), (5, 7)), ((8, 6, 10), (4, 4, 6))])
def test_variant_branches_match_verbatim(spatial, nm, kw):
    """separable / TT / resolution-changing / complex-data branches of the restatement, live against the
    verbatim module (the committed fixtures cover a fixed subset of these)."""
    ref = ref_verbatim.load_reference()
    torch.manual_seed(1)
    kw = dict(kw)
    if isinstance(kw.get("resolution_scaling_factor"), list):
        kw["resolution_scaling_factor"] = (kw["resolution_scaling_factor"] * 2)[:len(spatial)]
    cplx, sep = bool(kw.get("complex_data")), bool(kw.get("separable"))
    conv = ref.SpectralConv(3, 3, nm, **kw)
    with torch.no_grad():
        for prm in conv.weight.parameters():
            prm.copy_(torch.randn_like(prm) * 0.5)
    x = torch.randn(2, 3, *spatial, dtype=torch.cfloat if cplx else torch.float32)
    y = conv(x)
    y2 = so.forward_torch(x, conv.weight.to_tensor().detach(), conv.bias.detach(), conv.n_modes, conv.max_n_modes,
                          separable=sep, output_shape=list(y.shape[2:]), complex_data=cplx)
    assert y2.shape == y.shape and y2.dtype == y.dtype
    assert so.rel_l2(y2.detach().numpy(), y.detach().numpy()) < 2e-7


@pytest.mark.parametrize("fac", [None, "Tucker", "CP", "TT"])
@pytest.mark.parametrize("sep", [False, True])
def test_state_dict_layout_is_the_reference_containers(fac, sep):
    """Checkpoints move between the reference module and the drop-in: same parameter names and shapes
    (weight.tensor | weight.core + weight.factors.factor_i | weight.weights + weight.factors.factor_i, bias: tltorch's
    FactorList naming).  TT ranks are
    passed explicitly (the rank RULE is tensorly's and unpinned)."""
    from neuraloperator_amd import SpectralConv
    ref = ref_verbatim.load_reference()
    kw = dict(factorization=fac, separable=sep, rank=0.5)
    rconv = ref.SpectralConv(4, 4, (8, 6), **kw)
    if fac == "TT":
        kw["rank"] = [int(f.shape[0]) for f in rconv.weight.factors] + [1]
    mine = SpectralConv(4, 4, (8, 6), **kw)
    rs, ms = rconv.state_dict(), mine.state_dict()
    assert list(rs.keys()) == list(ms.keys())
    for k in rs:
        assert tuple(rs[k].shape) == tuple(ms[k].shape) and rs[k].dtype == ms[k].dtype, k
    mine.load_state_dict(rs)
    assert so.rel_l2(mine.weight.to_tensor().detach().numpy(), rconv.weight.to_tensor().detach().numpy()) < 1e-6


@pytest.mark.parametrize("shape", [(2, 5, 4, 8, 5), (3, 6, 6, 12)], ids=str)
def test_chalf_contraction_matches_verbatim_einsum_complexhalf(shape):
    """fno_block_precision half / mixed: oracle.contract_dense_chalf == the verbatim einsum_complexhalf
    (einsum_utils.py:10-83) on the dense equation the module builds (:21-46), bit for bit."""
    import sys
    ref_verbatim.load_reference()
    eu = sys.modules["neuralop.layers.einsum_utils"]
    b, ci, co = shape[:3]
    modes = shape[3:]
    g = torch.Generator().manual_seed(3)
    x = torch.randn(b, ci, *modes, dtype=torch.complex64, generator=g)
    w = torch.randn(ci, co, *modes, dtype=torch.complex64, generator=g)
    m = "cd"[:len(modes)]
    want = eu.einsum_complexhalf(f"ab{m},be{m}->ae{m}", x.chalf(), w.chalf())        # the module's call (:42-44)
    got = so.contract_dense_chalf(x, w)
    assert want.dtype == torch.complex32
    assert torch.equal(torch.view_as_real(want).float(), torch.view_as_real(got))
