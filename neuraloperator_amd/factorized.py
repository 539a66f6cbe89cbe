"""Parameter containers for the spectral weight (host glue).

The reference keeps its weight in a ``tltorch.FactorizedTensor``
(/root/reference/neuralop/layers/spectral_convolution.py:362-370) -- an un-vendored
third-party class.  These containers give the drop-in module the same surface the
reference's consumers touch:

* state-dict names ``weight.tensor`` (Dense), ``weight.core`` + ``weight.factors.factor_{i}``
  (Tucker), ``weight.weights`` + ``weight.factors.factor_{i}`` (CP), ``weight.factors.factor_{i}`` (TT) --
  tltorch's FactorList naming; checkpoints that store complex parameters as real (..., 2)
  views, or that use ``factors.{i}`` (this package, round 1), load as well
* ``.shape``, ``.name``, ``.normal_()``, ``.to_tensor()``, ``w[slices]`` (factor-row slicing
  for Tucker/CP), and use as a tensor in torch functions (``torch.zeros_like(w)``,
  ``w += ...``) as neuralop/training/incremental.py:215-238 does.

Rank selection follows tensorly's published rules (validate_tucker_rank / CP); no reference
test pins them ("parity unpinned"), so parity tests set factors
explicitly.
"""
import math
from typing import List, Optional, Sequence

import numpy as np
import torch
from torch import nn


def tucker_rank(shape: Sequence[int], rank, fixed_modes: Optional[List[int]] = None) -> List[int]:
    if isinstance(rank, (list, tuple)):
        return [int(r) for r in rank]
    if rank == "same":
        rank = 1.0
    if isinstance(rank, int) and not isinstance(rank, bool):
        return [min(int(rank), s) for s in shape]
    from scipy.optimize import brentq

    rank = float(rank)
    fixed = sorted(fixed_modes or [])
    comp = [s for i, s in enumerate(shape) if i not in fixed]
    n_fixed = int(np.prod([shape[i] for i in fixed])) if fixed else 1
    n_param = int(np.prod(comp)) * n_fixed
    sq = sum(s * s for s in comp)
    sq_fixed = sum(shape[i] ** 2 for i in fixed)          # factors of the fixed modes: full-rank squares
    n = len(comp)
    # tensorly.validate_tucker_rank: n_param x^n + (sum of squared compressed sizes) x + (fixed factors) x = rank n_param
    frac = brentq(lambda x: n_param * x ** n + sq * x + sq_fixed * x - rank * n_param, 0.0, max(rank, 1.0))
    comp_r = [max(int(round(s * frac)), 1) for s in comp]
    out, j = [], 0
    for i, s in enumerate(shape):
        if i in fixed:
            out.append(s)
        else:
            out.append(comp_r[j])
            j += 1
    return out


def cp_rank(shape: Sequence[int], rank) -> int:
    if isinstance(rank, int) and not isinstance(rank, bool):
        return int(rank)
    if rank == "same":
        rank = 1.0
    return max(int(round(float(rank) * np.prod(shape) / np.sum(shape))), 1)


def tt_rank(shape: Sequence[int], rank) -> List[int]:
    """Bond ranks (1, r_1, .., r_{n-1}, 1) of a tensor-train: tensorly's validate_tt_rank with its
    default constant_rank=False -- r_k proportional to the mean of the two neighbouring mode sizes,
    the proportion solving  sum(r_k s_k r_{k+1}) = rank * prod(shape)."""
    n = len(shape)
    if isinstance(rank, (list, tuple)):
        r = [int(x) for x in rank]
        if len(r) != n + 1 or r[0] != 1 or r[-1] != 1:
            raise ValueError(f"a TT rank for {n} modes has {n + 1} entries and starts/ends with 1, got {r}")
        return r
    if isinstance(rank, int) and not isinstance(rank, bool):
        return [1] + [int(rank)] * (n - 1) + [1]
    if rank == "same":
        rank = 1.0
    rank = float(rank)
    if n == 1:
        return [1, 1]
    avg = [(shape[i] + shape[i + 1]) / 2.0 for i in range(n - 1)]
    a = sum(avg[i - 1] * shape[i] * avg[i] for i in range(1, n - 1)) if n > 2 else avg[0] ** 2 * shape[0]
    b = shape[0] * avg[0] + shape[-1] * avg[-1]
    c = -float(np.prod(shape)) * rank
    frac = (-b + math.sqrt(b * b - 4 * a * c)) / (2 * a)
    return [1] + [max(int(round(d * frac)), 1) for d in avg] + [1]


def _as_param_dtype(value, param):
    """checkpoint tensor -> something ``param.copy_`` accepts: complex parameters may have been stored as real
    (..., 2) views"""
    if torch.is_tensor(value) and param.is_complex() and not value.is_complex() and value.shape[-1:] == (2,) \
            and tuple(value.shape[:-1]) == tuple(param.shape):
        return torch.view_as_complex(value.contiguous())
    return value


class FactorList(nn.Module):
    """List of factor parameters registered as ``factor_0, factor_1, ...`` (tltorch.FactorList: the state-dict names
    of a reference TFNO checkpoint are ``...weight.factors.factor_{i}``)."""

    def __init__(self, factors=()):
        super().__init__()
        self._n = 0
        for f in factors:
            self.append(f)

    def append(self, f):
        self.register_parameter(f"factor_{self._n}", f if isinstance(f, nn.Parameter) else nn.Parameter(f))
        self._n += 1
        return self

    def __len__(self):
        return self._n

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [self[j] for j in range(*i.indices(self._n))]
        if i < 0:
            i += self._n
        if not 0 <= i < self._n:
            raise IndexError(i)
        return getattr(self, f"factor_{i}")

    def __iter__(self):
        return (self[i] for i in range(self._n))

    def _load_from_state_dict(self, state_dict, prefix, *args, **kwargs):
        for i in range(self._n):
            new, old = f"{prefix}factor_{i}", f"{prefix}{i}"
            if new not in state_dict and old in state_dict:        # nn.ParameterList naming
                state_dict[new] = state_dict.pop(old)
            if new in state_dict:
                state_dict[new] = _as_param_dtype(state_dict[new], self[i])
        super()._load_from_state_dict(state_dict, prefix, *args, **kwargs)


class SpectralWeight(nn.Module):
    """Base container.  Sub-classes: DenseWeight, TuckerWeight, CPWeight, TTWeight."""

    name = "Base"

    def _load_from_state_dict(self, state_dict, prefix, *args, **kwargs):
        for pname, prm in self._parameters.items():
            key = prefix + pname
            if prm is not None and key in state_dict:
                state_dict[key] = _as_param_dtype(state_dict[key], prm)
        super()._load_from_state_dict(state_dict, prefix, *args, **kwargs)

    @staticmethod
    def new(shape, rank=1.0, factorization="Dense", fixed_rank_modes=None,
            dtype=torch.cfloat, device=None, **_ignored):
        f = (factorization or "Dense").lower()
        if f in ("dense", "complexdense"):
            return DenseWeight(torch.empty(*shape, dtype=dtype, device=device))
        if f in ("tucker", "complextucker"):
            r = tucker_rank(shape, rank, fixed_rank_modes)
            return TuckerWeight(torch.empty(*r, dtype=dtype, device=device),
                                [torch.empty(s, ri, dtype=dtype, device=device)
                                 for s, ri in zip(shape, r)])
        if f in ("cp", "complexcp"):
            r = cp_rank(shape, rank)
            return CPWeight(torch.ones(r, dtype=dtype, device=device),
                            [torch.empty(s, r, dtype=dtype, device=device) for s in shape])
        if f in ("tt", "complextt"):
            r = tt_rank(shape, rank)
            return TTWeight([torch.empty(r[i], s, r[i + 1], dtype=dtype, device=device)
                             for i, s in enumerate(shape)])
        raise ValueError(f"factorization={factorization!r}: expected Dense, Tucker, CP or TT")

    # tensor-like behaviour for host code (incremental trainer, regularisers)
    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}

        def unwrap(a):
            if isinstance(a, SpectralWeight):
                return a.to_tensor()
            if isinstance(a, (list, tuple)):
                return type(a)(unwrap(b) for b in a)
            return a

        return func(*unwrap(args), **{k: unwrap(v) for k, v in kwargs.items()})

    @property
    def dtype(self):
        return next(self.parameters()).dtype

    @property
    def device(self):
        return next(self.parameters()).device

    def size(self, dim=None):
        return self.shape if dim is None else self.shape[dim]

    @property
    def ndim(self):
        return len(self.shape)


class DenseWeight(SpectralWeight):
    name = "Dense"

    def __init__(self, tensor):
        super().__init__()
        self.tensor = nn.Parameter(tensor)

    @property
    def shape(self):
        return self.tensor.shape

    def normal_(self, mean=0.0, std=1.0):
        with torch.no_grad():
            self.tensor.normal_(mean, std)
        return self

    def to_tensor(self):
        return self.tensor

    def __getitem__(self, idx):
        return self.tensor[idx]


def _slices(idx, n):
    if not isinstance(idx, tuple):
        idx = (idx,)
    idx = tuple(idx) + (slice(None),) * (n - len(idx))
    if not all(isinstance(s, slice) for s in idx):
        raise IndexError("factorized spectral weights support slice indexing only")
    return idx


class TuckerWeight(SpectralWeight):
    name = "Tucker"

    def __init__(self, core, factors, as_parameters=True):
        super().__init__()
        if as_parameters:
            self.core = nn.Parameter(core)
            self.factors = FactorList(factors)
        else:  # a sliced view: shares storage with the parent's parameters
            self.core = core
            self.factors = list(factors)

    @property
    def shape(self):
        return torch.Size([f.shape[0] for f in self.factors])

    @property
    def rank(self):
        return tuple(self.core.shape)

    def normal_(self, mean=0.0, std=1.0):
        r = float(np.prod([math.sqrt(x) for x in self.core.shape]))
        std_f = (std / r) ** (1.0 / (len(self.factors) + 1))
        with torch.no_grad():
            self.core.normal_(0, std_f)
            for f in self.factors:
                f.normal_(0, std_f)
        return self

    def to_tensor(self):
        res = self.core
        for d, f in enumerate(self.factors):
            res = torch.movedim(torch.tensordot(f, res, dims=([1], [d])), 0, d)
        return res

    def __getitem__(self, idx):
        idx = _slices(idx, len(self.factors))
        return TuckerWeight(self.core, [f[s, :] for f, s in zip(self.factors, idx)],
                            as_parameters=False)


class CPWeight(SpectralWeight):
    name = "CP"

    def __init__(self, weights, factors, as_parameters=True):
        super().__init__()
        if as_parameters:
            self.weights = nn.Parameter(weights)
            self.factors = FactorList(factors)
        else:
            self.weights = weights
            self.factors = list(factors)

    @property
    def shape(self):
        return torch.Size([f.shape[0] for f in self.factors])

    @property
    def rank(self):
        return self.weights.shape[0]

    def normal_(self, mean=0.0, std=1.0):
        std_f = (std / math.sqrt(self.rank)) ** (1.0 / len(self.factors))
        with torch.no_grad():
            self.weights.fill_(1)
            for f in self.factors:
                f.normal_(0, std_f)
        return self

    def to_tensor(self):
        res = self.weights
        for f in self.factors:                      # (..., r) x (s, r) -> (..., s, r)
            res = res.unsqueeze(-2) * f
        return res.sum(-1)

    def __getitem__(self, idx):
        idx = _slices(idx, len(self.factors))
        return CPWeight(self.weights, [f[s, :] for f, s in zip(self.factors, idx)],
                        as_parameters=False)


class TTWeight(SpectralWeight):
    """Tensor-train cores G_k of shape (r_k, s_k, r_{k+1}), r_0 = r_n = 1
    (the operand layout of the reference's ``_contract_tt``, spectral_convolution.py:106-132)."""

    name = "TT"

    def __init__(self, factors, as_parameters=True):
        super().__init__()
        if as_parameters:
            self.factors = FactorList(factors)
        else:
            self.factors = list(factors)

    @property
    def shape(self):
        return torch.Size([f.shape[1] for f in self.factors])

    @property
    def rank(self):
        return tuple(int(f.shape[0]) for f in self.factors) + (1,)

    def normal_(self, mean=0.0, std=1.0):
        r = float(np.prod(self.rank))
        std_f = (std / r) ** (1.0 / len(self.factors))
        with torch.no_grad():
            for f in self.factors:
                f.normal_(0, std_f)
        return self

    def to_tensor(self):
        res = self.factors[0]                               # (1, s_0, r_1)
        for f in self.factors[1:]:
            res = torch.tensordot(res, f, dims=([res.ndim - 1], [0]))
        return res.squeeze(0).squeeze(-1)

    def __getitem__(self, idx):
        idx = _slices(idx, len(self.factors))
        return TTWeight([f[:, s, :] for f, s in zip(self.factors, idx)], as_parameters=False)
