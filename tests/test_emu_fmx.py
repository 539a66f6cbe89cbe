"""CPU tier: contractions with a mode-independent factor matrix and mode-summed contractions on the matrix cores
(sc_kernels_fmx.h: k_modegemm_bfac_mx, k_modegemm_msum_mx + k_fmx_reduce) in host emulation against numpy complex128:
the factor steps of the Tucker / CP chains and the gradients of the factors (spectral_convolution.py:55-103 and its
autograd), GaLore's mode products.  Ragged ranks (36, 19), mode counts that do not fill the 64-mode chunks, every
conjugation, both storage orders of the factor, several chunks per workgroup, the dispatch rule."""
import numpy as np
import pytest
import torch

from engine_runner import emu_lib, rel_l2
from neuraloperator_amd import _lib

TOL = 2e-6


@pytest.fixture(scope="module")
def lib():
    return emu_lib()


def _rand(*shape, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.complex(torch.randn(*shape, generator=g), torch.randn(*shape, generator=g))


def _p(t):
    return torch.view_as_real(t).data_ptr()


@pytest.mark.parametrize("dims", [(3, 64, 36, 130), (2, 36, 64, 70), (5, 10, 19, 64), (1, 5, 8, 200), (2, 7, 40, 66),
                                  (300, 4, 9, 64)], ids=lambda d: "P%d_R%d_Q%d_M%d" % d)
@pytest.mark.parametrize("transposed", [False, True], ids=["B_rq", "B_qr"])
@pytest.mark.parametrize("conj", [(0, 0), (0, 1), (1, 0), (1, 1)], ids=["plain", "conjB", "conjA", "conjAB"])
def test_factor_operand_on_matrix_cores(lib, dims, transposed, conj):
    P, R, Q, M = dims
    ca, cb = conj
    a = _rand(P, R, M, seed=7)
    bm = _rand(R, Q, seed=8)
    store = bm.t().contiguous() if transposed else bm.contiguous()
    kw = dict(P=P, Q=Q, R=R, n_modes=M, a_sp=R * M, a_sr=M, a_sm=1, b_sm=0, conj_a=ca, conj_b=cb,
              b_sr=(1 if transposed else Q), b_sq=(R if transposed else 1), c_sp=Q * M, c_sq=M, c_sm=1)
    assert lib.modegemm_path(**kw) == 5
    a128, b128 = a.numpy().astype(np.complex128), bm.numpy().astype(np.complex128)
    ref = np.einsum("prm,rq->pqm", np.conj(a128) if ca else a128, np.conj(b128) if cb else b128)
    c = torch.full((P, Q, M), float("nan"), dtype=torch.complex64)
    lib.modegemm(_p(a), _p(store), _p(c), 0, **kw)
    assert rel_l2(c.numpy(), ref) < TOL


def test_factor_operand_strided_rows(lib):
    """x[b, i, m] contracted over i with U[i, f], output rows f strided like the chain's (B, F, M) tensor; the
    operand a slice of a larger array."""
    B, Ci, F, M = 3, 12, 9, 96
    big = _rand(B, Ci + 2, M + 8, seed=3)
    a = big[:, 1:Ci + 1, 4:M + 4]
    u = _rand(Ci, F, seed=4)
    kw = dict(P=B, Q=F, R=Ci, n_modes=M, a_sp=a.stride(0), a_sr=a.stride(1), a_sm=1, b_sr=F, b_sq=1, b_sm=0,
              c_sp=F * M, c_sq=M, c_sm=1)
    assert lib.modegemm_path(**kw) == 5
    c = torch.full((B, F, M), float("nan"), dtype=torch.complex64)
    lib.modegemm(_p(a), _p(u), _p(c), 0, **kw)
    ref = np.einsum("bim,if->bfm", a.numpy().astype(np.complex128), u.numpy().astype(np.complex128))
    assert rel_l2(c.numpy(), ref) < TOL


def test_dispatch_rule(lib):
    base = dict(P=4, Q=36, R=64, n_modes=128, a_sp=64 * 128, a_sr=128, a_sm=1, b_sr=36, b_sq=1, b_sm=0, c_sp=36 * 128,
                c_sq=128, c_sm=1)
    assert lib.modegemm_path(**base) == 5
    assert lib.modegemm_path(flags=_lib.SC_GEMM_NO_FMX, **base) == 4
    assert lib.modegemm_path(**{**base, "Q": 65, "c_sp": 65 * 128, "b_sr": 65}) == 4      # wider than the tile set
    assert lib.modegemm_path(**{**base, "R": 65, "a_sp": 65 * 128}) == 4
    assert lib.modegemm_path(**{**base, "n_modes": 32}) != 5                                 # too few modes
    assert lib.modegemm_path(**{**base, "b_sm": 1}) != 5                                     # a per-mode operand


# (P, R, Q, M)
@pytest.mark.parametrize("dims", [(64, 5, 36, 130), (36, 3, 64, 200), (20, 2, 50, 70), (8, 40, 8, 64), (64, 2, 64, 1300)],
                         ids=lambda d: "P%d_R%d_Q%d_M%d" % d)
@pytest.mark.parametrize("conj", [(1, 0), (0, 1), (0, 0), (1, 1)], ids=["conjA", "conjB", "plain", "conjAB"])
def test_mode_summed_contraction(lib, dims, conj):
    """C[p, q] = sum_{r, m} opA(A[p, r, m]) opB(B[r, q, m]): the gradient of a factor matrix -- A given as the
    transposed view of an (R, P, M) activation tensor, as the autograd of the chain passes it."""
    P, R, Q, M = dims
    ca, cb = conj
    act = _rand(R, P, M, seed=11)                       # e.g. xhat[b, i, m]: rows r = b, p = i
    a = act.transpose(0, 1)                              # [P, R, M] view
    b = _rand(R, Q, M, seed=12)
    kw = dict(P=P, Q=Q, R=R, n_modes=M, a_sp=a.stride(0), a_sr=a.stride(1), a_sm=1, b_sr=Q * M, b_sq=M, b_sm=1,
              conj_a=ca, conj_b=cb, c_sp=Q, c_sq=1, c_sm=0)
    nbytes = lib.modegemm_msum_workspace_bytes(**kw)
    assert nbytes > 0
    ws = torch.empty(nbytes, dtype=torch.uint8)
    c = torch.full((P, Q), float("nan"), dtype=torch.complex64)
    lib.modegemm_msum_ws(_p(act), _p(b), _p(c), ws.data_ptr(), nbytes, 0, **kw)
    a128, b128 = a.numpy().astype(np.complex128), b.numpy().astype(np.complex128)
    ref = np.einsum("prm,rqm->pq", np.conj(a128) if ca else a128, np.conj(b128) if cb else b128)
    assert rel_l2(c.numpy(), ref) < TOL
    # the atomic-add kernel it replaces for these shapes
    c0 = torch.zeros((P, Q), dtype=torch.complex64)
    lib.modegemm_msum(_p(act), _p(b), _p(c0), 0, **kw)
    assert rel_l2(c0.numpy(), ref) < 1e-5


def test_mode_summed_limits(lib):
    kw = dict(P=64, Q=36, R=4, n_modes=128, a_sp=128, a_sr=64 * 128, a_sm=1, b_sr=36 * 128, b_sq=128, b_sm=1, c_sp=36,
              c_sq=1, c_sm=0)
    assert lib.modegemm_msum_workspace_bytes(**kw) > 0 and lib.modegemm_msum_path(**kw) == 1
    # outside the matrix-core kernel's shapes: the slot form of the VALU kernel (still a workspace, still no atomics)
    for other in ({**kw, "P": 65}, {**kw, "Q": 4}, {**kw, "b_sm": 0}, {**kw, "flags": _lib.SC_GEMM_NO_FMX}):
        assert lib.modegemm_msum_path(**other) == 0 and lib.modegemm_msum_workspace_bytes(**other) > 0
    with pytest.raises(Exception):
        lib.modegemm_msum_ws(0, 0, 0, 0, 0, 0, **kw)


@pytest.mark.parametrize("dims", [(6, 3, 5, 150), (10, 4, 7, 64), (19, 2, 33, 60), (65, 3, 36, 130)], ids=str)
@pytest.mark.parametrize("conj", [(1, 0), (0, 0)], ids=["conjA", "plain"])
def test_mode_summed_contraction_slot_form(lib, dims, conj):
    """Shapes the matrix-core kernel refuses (small Tucker / CP factors, P > 64): sc_modegemm_msum_ws runs the VALU
    kernel with one workspace slot per (mode split, r split) and the fixed-order reduction -- no atomics."""
    P, R, Q, M = dims
    ca, cb = conj
    act = _rand(R, P, M, seed=21)
    a = act.transpose(0, 1)
    b = _rand(R, Q, M, seed=22)
    kw = dict(P=P, Q=Q, R=R, n_modes=M, a_sp=a.stride(0), a_sr=a.stride(1), a_sm=1, b_sr=Q * M, b_sq=M, b_sm=1,
              conj_a=ca, conj_b=cb, c_sp=Q, c_sq=1, c_sm=0)
    assert lib.modegemm_msum_path(**kw) == 0
    nbytes = lib.modegemm_msum_workspace_bytes(**kw)
    assert nbytes > 0
    ws = torch.full((nbytes,), 0xFF, dtype=torch.uint8)                 # NaN patterns: every slot must be written
    c = torch.full((P, Q), float("nan"), dtype=torch.complex64)
    lib.modegemm_msum_ws(_p(act), _p(b), _p(c), ws.data_ptr(), nbytes, 0, **kw)
    a128, b128 = a.numpy().astype(np.complex128), b.numpy().astype(np.complex128)
    ref = np.einsum("prm,rqm->pq", np.conj(a128) if ca else a128, np.conj(b128) if cb else b128)
    assert rel_l2(c.numpy(), ref) < 1e-5
