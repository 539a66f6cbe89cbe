"""CPU tier: the spectral-weight containers (neuraloperator_amd/factorized.py) -- the surface of
tltorch.FactorizedTensor the reference's consumers touch: state-dict names, checkpoint
loading (tltorch names, round-1 names, complex parameters stored as real (..., 2) views), rank rules, slicing,
tensor-like use by the incremental trainer (neuralop/training/incremental.py:215-238)."""
import pytest
import torch

from neuraloperator_amd.factorized import SpectralWeight, tucker_rank

SHAPE = (6, 5, 8, 4)


def _new(fac, rank=0.5):
    w = SpectralWeight.new(SHAPE, rank=rank, factorization=fac)
    w.normal_(0, 0.3)
    return w


def test_state_dict_names_follow_tltorch():
    assert list(_new("Dense").state_dict()) == ["tensor"]
    assert list(_new("Tucker").state_dict()) == ["core"] + [f"factors.factor_{i}" for i in range(4)]
    assert list(_new("CP").state_dict()) == ["weights"] + [f"factors.factor_{i}" for i in range(4)]
    assert list(_new("TT").state_dict()) == [f"factors.factor_{i}" for i in range(4)]


@pytest.mark.parametrize("fac", ["Dense", "Tucker", "CP", "TT"])
def test_checkpoint_variants_load(fac):
    src, dst = _new(fac), _new(fac)
    sd = src.state_dict()
    # (a) as saved; (b) round-1 nn.ParameterList names; (c) complex parameters as real (..., 2) views
    legacy = {k.replace("factors.factor_", "factors."): v for k, v in sd.items()}
    as_real = {k: torch.view_as_real(v).clone() if v.is_complex() else v for k, v in sd.items()}
    for variant in (sd, legacy, as_real):
        for p in dst.parameters():
            p.data.zero_()
        dst.load_state_dict(dict(variant))
        assert torch.equal(dst.to_tensor(), src.to_tensor())
    with pytest.raises(RuntimeError):
        dst.load_state_dict({"nonsense": torch.zeros(1)})


def test_factor_list_behaves_like_a_list():
    w = _new("Tucker")
    assert len(w.factors) == 4 and w.factors[-1] is w.factors[3]
    assert [tuple(f.shape) for f in w.factors] == [(s, r) for s, r in zip(SHAPE, w.rank)]
    assert len(list(w.parameters())) == 5 and len(w.factors[1:3]) == 2
    with pytest.raises(IndexError):
        w.factors[4]


def test_tucker_rank_rule():
    # tensorly.validate_tucker_rank: SURVEY.md 8 row a6 pins (64, 64, 64, 33) at rank 0.1 -> (36, 36, 36, 19)
    assert tucker_rank((64, 64, 64, 33), 0.1) == [36, 36, 36, 19]
    # fixed modes keep their full size and their factors (size^2 parameters each) stay in the budget equation
    r = tucker_rank((64, 64, 64, 33), 0.1, fixed_modes=[0])
    assert r[0] == 64 and r[1:] != [36, 36, 19]
    full = 64 * 64 * 64 * 33
    n_par = r[0] * r[1] * r[2] * r[3] + sum(s * k for s, k in zip((64, 64, 64, 33), r))
    assert abs(n_par / full - 0.1) < 0.01
    assert tucker_rank((8, 8), [3, 2]) == [3, 2] and tucker_rank((8, 8), 5) == [5, 5]


@pytest.mark.parametrize("fac", ["Dense", "Tucker", "CP", "TT"])
def test_slicing_and_tensor_like_use(fac):
    w = _new(fac)
    sl = (slice(None), slice(None), slice(1, 7), slice(None, 3))
    sub = w[sl]
    assert tuple(sub.shape) == (6, 5, 6, 3)
    dense = sub if torch.is_tensor(sub) else sub.to_tensor()
    assert torch.allclose(dense, w.to_tensor()[sl], atol=1e-6)
    # incremental trainer: torch.zeros_like(weight), accumulating weight slices
    acc = torch.zeros_like(w)
    acc = acc + w
    assert tuple(acc.shape) == SHAPE and torch.allclose(acc, w.to_tensor())
