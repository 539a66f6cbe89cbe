"""TEST INFRASTRUCTURE ONLY -- stand-ins for the reference's un-vendored third-party deps.

The reference's hot-path file imports ``tensorly``, ``tensorly.plugins``,
``tltorch.factorized_tensors.core`` and (for the chalf path) ``opt_einsum``
(/root/reference/neuralop/layers/spectral_convolution.py:8-12,
einsum_utils.py:2-7).  None is installed in this image and none is vendored
under /root/reference, so the verbatim reference can only be imported through
stubs.  What the stubs restate (from the packages' published behaviour; see
SURVEY.md section 8c "[3p-memory]"):

* ``tl.einsum``  -> pairwise torch.einsum along a greedy smallest-intermediate
  path (opt_einsum's job; a left-to-right multi-operand torch.einsum would
  build outer products, SURVEY.md section 8c "Oracle caveat").
* ``FactorizedTensor.new(shape, rank, factorization, ...)`` for Dense /
  Tucker / CP / TT, ``.normal_``, ``__getitem__`` with a tuple of slices
  (factor-row slicing for Tucker/CP/TT), ``.shape``, ``.to_tensor()``,
  ``.name``, ``.core/.factors/.weights``.

Rank selection / init statistics follow tensorly's documented rules but are
"parity unpinned" (no reference test fixes them).
"""
import math
import sys
import types

import numpy as np
import torch
from torch import nn


# --------------------------------------------------------------------------
# einsum with an explicit pairwise path
# --------------------------------------------------------------------------
def _pairwise_einsum(eq, *ops):
    eq = eq.replace(" ", "")
    lhs, out = eq.split("->")
    terms = lhs.split(",")
    assert len(terms) == len(ops)
    ops = list(ops)
    if len(ops) <= 2:
        return torch.einsum(eq, *ops)
    sizes = {}
    for t, o in zip(terms, ops):
        for s, n in zip(t, o.shape):
            sizes[s] = n
    while len(ops) > 2:
        best = None
        for a in range(len(ops)):
            for b in range(a + 1, len(ops)):
                others = "".join(terms[c] for c in range(len(ops)) if c not in (a, b)) + out
                keep = [s for s in dict.fromkeys(terms[a] + terms[b]) if s in others]
                shared = set(terms[a]) & set(terms[b])
                size = int(np.prod([sizes[s] for s in keep])) if keep else 1
                # never pick a pure outer product unless nothing else is left
                cost = (0 if shared else 1, size)
                if best is None or cost < best[0]:
                    best = (cost, a, b, "".join(keep))
        _, a, b, keep = best
        res = torch.einsum(f"{terms[a]},{terms[b]}->{keep}", ops[a], ops[b])
        for idx in (b, a):
            ops.pop(idx)
            terms.pop(idx)
        ops.append(res)
        terms.append(keep)
    return torch.einsum(",".join(terms) + "->" + out, *ops)


# --------------------------------------------------------------------------
# rank rules (tensorly.tucker_tensor.validate_tucker_rank / cp / tt)
# --------------------------------------------------------------------------
def tucker_rank(shape, rank, fixed_modes=None):
    from scipy.optimize import brentq

    shape = list(shape)
    if isinstance(rank, (list, tuple)):
        return [int(r) for r in rank]
    if rank == "same":
        rank = 1.0
    rank = float(rank)
    fixed = sorted(fixed_modes or [])
    comp = [s for i, s in enumerate(shape) if i not in fixed]
    n_fixed = int(np.prod([shape[i] for i in fixed])) if fixed else 1
    n_param = int(np.prod(comp)) * n_fixed
    sq = sum(s * s for s in comp)
    n = len(comp)
    fun = lambda x: n_param * x ** n + sq * x - rank * n_param
    frac = brentq(fun, 0.0, max(rank, 1.0))
    out = [max(int(round(s * frac)), 1) for s in comp]
    res, j = [], 0
    for i, s in enumerate(shape):
        if i in fixed:
            res.append(s)
        else:
            res.append(out[j])
            j += 1
    return res


def cp_rank(shape, rank):
    if isinstance(rank, int) and not isinstance(rank, bool):
        return rank
    if rank == "same":
        rank = 1.0
    return max(int(round(float(rank) * np.prod(shape) / np.sum(shape))), 1)


def tt_rank(shape, rank):
    from scipy.optimize import brentq  # noqa: F401

    if isinstance(rank, (list, tuple)):
        return [int(r) for r in rank]
    if rank == "same":
        rank = 1.0
    rank = float(rank)
    n = len(shape)
    # solve a*r^2 + b*r - target = 0 for a constant interior rank r
    a = sum(shape[1:-1])
    b = shape[0] + shape[-1]
    target = rank * np.prod(shape)
    if a == 0:
        r = target / b
    else:
        r = (-b + math.sqrt(b * b + 4 * a * target)) / (2 * a)
    r = max(int(round(r)), 1)
    return [1] + [r] * (n - 1) + [1]


# --------------------------------------------------------------------------
# FactorizedTensor stand-ins
# --------------------------------------------------------------------------
class FactorizedTensor(nn.Module):
    name = "Base"

    @classmethod
    def new(cls, shape, rank=1.0, factorization="Dense", fixed_rank_modes=None,
            dtype=torch.cfloat, device=None, **kw):
        f = (factorization or "Dense").lower()
        if f.endswith("dense"):
            return DenseTensor(torch.empty(*shape, dtype=dtype, device=device))
        if f.endswith("tucker"):
            r = tucker_rank(shape, rank, fixed_rank_modes)
            core = torch.empty(*r, dtype=dtype, device=device)
            facs = [torch.empty(s, ri, dtype=dtype, device=device) for s, ri in zip(shape, r)]
            return TuckerTensor(core, facs)
        if f.endswith("cp"):
            r = cp_rank(shape, rank)
            w = torch.ones(r, dtype=dtype, device=device)
            facs = [torch.empty(s, r, dtype=dtype, device=device) for s in shape]
            return CPTensor(w, facs)
        if f.endswith("tt"):
            r = tt_rank(shape, rank)
            facs = [torch.empty(r[i], s, r[i + 1], dtype=dtype, device=device)
                    for i, s in enumerate(shape)]
            return TTTensor(facs)
        raise ValueError(factorization)

    @classmethod
    def from_tensor(cls, tensor, rank=None, factorization="Dense", **kw):
        assert factorization.lower().endswith("dense")
        return DenseTensor(tensor.detach().clone())

    def __torch_function__(self, func, types_, args=(), kwargs=None):  # pragma: no cover
        return NotImplemented


class DenseTensor(FactorizedTensor):
    name = "Dense"

    def __init__(self, tensor):
        super().__init__()
        self.tensor = nn.Parameter(tensor)

    @property
    def shape(self):
        return self.tensor.shape

    def normal_(self, mean=0, std=1):
        with torch.no_grad():
            self.tensor.normal_(mean, std)
        return self

    def to_tensor(self):
        return self.tensor

    def __getitem__(self, idx):
        return self.tensor[idx]


def _as_slices(idx, n):
    if not isinstance(idx, tuple):
        idx = (idx,)
    idx = tuple(idx) + (slice(None),) * (n - len(idx))
    assert all(isinstance(s, slice) for s in idx), "stub supports slice indexing only"
    return idx


class FactorList(nn.Module):
    """tltorch.FactorList: the factors of a decomposition as parameters named factor_0, factor_1, ... (state-dict
    keys ``...factors.factor_{i}``)"""

    def __init__(self, factors=()):
        super().__init__()
        self._n = 0
        for f in factors:
            self.register_parameter(f"factor_{self._n}", f if isinstance(f, nn.Parameter) else nn.Parameter(f))
            self._n += 1

    def __len__(self):
        return self._n

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [self[j] for j in range(*i.indices(self._n))]
        return getattr(self, f"factor_{i % self._n}")

    def __iter__(self):
        return (self[i] for i in range(self._n))


class TuckerTensor(FactorizedTensor):
    name = "Tucker"

    def __init__(self, core, factors, _param=True):
        super().__init__()
        if _param:
            self.core = nn.Parameter(core)
            self.factors = FactorList(factors)
        else:
            self.core = core
            self.factors = list(factors)

    @property
    def shape(self):
        return torch.Size([f.shape[0] for f in self.factors])

    @property
    def rank(self):
        return tuple(self.core.shape)

    def normal_(self, mean=0, std=1):
        r = np.prod([math.sqrt(x) for x in self.core.shape])
        std_f = (std / r) ** (1 / (len(self.factors) + 1))
        with torch.no_grad():
            self.core.normal_(0, std_f)
            for f in self.factors:
                f.normal_(0, std_f)
        return self

    def to_tensor(self):
        n = len(self.factors)
        sy = "abcdefghijklmnopqrstuvwxyz"
        core_s, out_s = sy[:n], sy[n:2 * n]
        eq = core_s + "," + ",".join(o + c for o, c in zip(out_s, core_s)) + "->" + out_s
        return _pairwise_einsum(eq, self.core, *self.factors)

    def __getitem__(self, idx):
        idx = _as_slices(idx, len(self.factors))
        return TuckerTensor(self.core, [f[s, :] for f, s in zip(self.factors, idx)], _param=False)


class CPTensor(FactorizedTensor):
    name = "CP"

    def __init__(self, weights, factors, _param=True):
        super().__init__()
        if _param:
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

    def normal_(self, mean=0, std=1):
        std_f = (std / math.sqrt(self.rank)) ** (1 / len(self.factors))
        with torch.no_grad():
            self.weights.fill_(1)
            for f in self.factors:
                f.normal_(0, std_f)
        return self

    def to_tensor(self):
        n = len(self.factors)
        sy = "abcdefghijklmnopqrstuvwxyz"
        out_s = sy[:n]
        eq = "z," + ",".join(o + "z" for o in out_s) + "->" + out_s
        return _pairwise_einsum(eq, self.weights, *self.factors)

    def __getitem__(self, idx):
        idx = _as_slices(idx, len(self.factors))
        return CPTensor(self.weights, [f[s, :] for f, s in zip(self.factors, idx)], _param=False)


class TTTensor(FactorizedTensor):
    name = "TT"

    def __init__(self, factors, _param=True):
        super().__init__()
        if _param:
            self.factors = FactorList(factors)
        else:
            self.factors = list(factors)

    @property
    def shape(self):
        return torch.Size([f.shape[1] for f in self.factors])

    def normal_(self, mean=0, std=1):
        r = np.prod([f.shape[0] for f in self.factors])
        std_f = (std / r) ** (1 / len(self.factors))
        with torch.no_grad():
            for f in self.factors:
                f.normal_(0, std_f)
        return self

    def to_tensor(self):
        res = self.factors[0]  # (1, s0, r1)
        for f in self.factors[1:]:
            res = torch.einsum("...a,abc->...bc", res, f)
        return res.squeeze(0).squeeze(-1)

    def __getitem__(self, idx):
        idx = _as_slices(idx, len(self.factors))
        return TTTensor([f[:, s, :] for f, s in zip(self.factors, idx)], _param=False)


# --------------------------------------------------------------------------
# sys.modules installation
# --------------------------------------------------------------------------
def install():
    """Register the stub modules (idempotent)."""
    if "tltorch.factorized_tensors.core" in sys.modules and getattr(
            sys.modules["tltorch.factorized_tensors.core"], "_IS_ORACLE_STUB", False):
        return
    tl = types.ModuleType("tensorly")
    tl.set_backend = lambda *_a, **_k: None
    tl.ndim = lambda t: t.ndim
    tl.einsum = _pairwise_einsum
    tl._IS_ORACLE_STUB = True
    plugins = types.ModuleType("tensorly.plugins")
    plugins.use_opt_einsum = lambda *_a, **_k: None
    tl.plugins = plugins

    oe = types.ModuleType("opt_einsum")

    def _contract_path(eq, *ops):  # only used by the chalf path
        n = len(ops)
        path = [(0, 1)] * (n - 1)
        return path, None

    oe.contract_path = _contract_path
    oe.contract = _pairwise_einsum

    tlt = types.ModuleType("tltorch")
    tlt.FactorizedTensor = FactorizedTensor
    ft = types.ModuleType("tltorch.factorized_tensors")
    core = types.ModuleType("tltorch.factorized_tensors.core")
    core.FactorizedTensor = FactorizedTensor
    core._IS_ORACLE_STUB = True
    ft.core = core
    tlt.factorized_tensors = ft

    sys.modules.setdefault("tensorly", tl)
    sys.modules.setdefault("tensorly.plugins", plugins)
    sys.modules.setdefault("opt_einsum", oe)
    sys.modules.setdefault("tltorch", tlt)
    sys.modules.setdefault("tltorch.factorized_tensors", ft)
    sys.modules.setdefault("tltorch.factorized_tensors.core", core)
