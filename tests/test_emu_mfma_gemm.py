"""CPU tier: the matrix-core mode GEMM (sc_kernels_mfma.h) in host emulation -- the MFMA is
replaced by its documented lane/register map (cdna_hip_programming.md section 3), everything
else (mode ranges, LDS staging, operand sign masks, C scatter) is the product source -- against
a numpy complex128 einsum.  Covers the three contractions of the layer (forward, gX, gW with
their conjugations / transposed strides), ranges of 1, 8 and 9 modes, r tails, sub-block
index tables, and agreement with the lanes-are-modes VALU kernel."""
import numpy as np
import pytest
import torch

from engine_runner import emu_lib, rel_l2
from neuraloperator_amd import _lib

TOL = 1e-5


@pytest.fixture(scope="module")
def lib():
    return emu_lib()


def _c(t):
    return torch.view_as_real(t.contiguous())


def _rand(*shape, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.complex(torch.randn(*shape, generator=g), torch.randn(*shape, generator=g))


def run_gemm(lib, a, b, c, flags=0, **kw):
    av, bv, cv = _c(a), _c(b), torch.view_as_real(c)
    on_mfma = lib.modegemm_uses_matrix_cores(flags=flags, **kw)
    assert on_mfma == (not (flags & _lib.SC_GEMM_FORCE_VALU)), "test must exercise the intended kernel"
    lib.modegemm(av.data_ptr(), bv.data_ptr(), cv.data_ptr(), 0, flags=flags, **kw)
    return c


# (B, Ci, Co, M, grid cap): grid cap 0 = auto (one mode per workgroup below 256 modes)
CASES = [
    (32, 64, 64, 5, 0),      # nm = 1 everywhere
    (32, 64, 64, 26, 3),     # ranges of 8, 9, 9 modes
    (32, 20, 64, 17, 2),     # r tail (20 = 8 + 8 + 4), ranges 8 and 9
    (64, 64, 64, 9, 1),      # P = 64 for forward (8 waves, 9 modes)
    (64, 16, 64, 17, 2),     # wide shape, ranges 8 and 9
    (32, 64, 36, 17, 2),     # ragged columns (a Tucker rank): 36 of the tile's 64
    (28, 36, 36, 10, 2),     # ragged rows, columns and r
    (36, 32, 40, 9, 1),      # 36 rows on the 64-row shape
]


@pytest.mark.parametrize("shape", [_lib.SC_GEMM_WIDE, 0], ids=["wide9", "few4"])
@pytest.mark.parametrize("case", CASES, ids=lambda c: "B%d_Ci%d_Co%d_M%d_g%d" % c)
def test_forward_contraction(lib, case, shape):
    """shape: 9-mode workgroups (forced: the fixtures have few modes) / the 4-mode shape small problems take"""
    B, Ci, Co, M, cap = case
    x = _rand(B, Ci, M, seed=1)
    w = _rand(Ci, Co, M, seed=2)
    y = torch.zeros(B, Co, M, dtype=torch.complex64)
    run_gemm(lib, x, w, y, flags=shape | _lib.SC_GEMM_GRID(cap), P=B, Q=Co, R=Ci, n_modes=M,
             a_sp=Ci * M, a_sr=M, a_sm=1, b_sr=Co * M, b_sq=M, b_sm=1, c_sp=Co * M, c_sq=M, c_sm=1)
    ref = np.einsum("bim,iom->bom", x.numpy().astype(np.complex128), w.numpy().astype(np.complex128))
    assert rel_l2(y.numpy(), ref) < TOL
    if B <= 32:   # the paired (4-wave, <= 5 modes) shape of the same contraction
        y3 = torch.zeros_like(y)
        wcap = max(1, -(-M // 5))
        run_gemm(lib, x, w, y3, flags=_lib.SC_GEMM_PAIRED | _lib.SC_GEMM_GRID(wcap), P=B, Q=Co, R=Ci, n_modes=M,
                 a_sp=Ci * M, a_sr=M, a_sm=1, b_sr=Co * M, b_sq=M, b_sm=1, c_sp=Co * M, c_sq=M, c_sm=1)
        assert rel_l2(y3.numpy(), ref) < TOL
    # same call on the VALU kernel
    y2 = torch.zeros_like(y)
    run_gemm(lib, x, w, y2, flags=_lib.SC_GEMM_FORCE_VALU, P=B, Q=Co, R=Ci, n_modes=M,
             a_sp=Ci * M, a_sr=M, a_sm=1, b_sr=Co * M, b_sq=M, b_sm=1, c_sp=Co * M, c_sq=M, c_sm=1)
    assert rel_l2(y2.numpy(), ref) < TOL


def test_gx_contraction_conj_b_transposed_strides(lib):
    B, Ci, Co, M = 32, 64, 16, 19
    g = _rand(B, Co, M, seed=3)
    w = _rand(Ci, Co, M, seed=4)
    gx = torch.zeros(B, Ci, M, dtype=torch.complex64)
    run_gemm(lib, g, w, gx, flags=_lib.SC_GEMM_WIDE | _lib.SC_GEMM_GRID(3), P=B, Q=Ci, R=Co, n_modes=M,
             a_sp=Co * M, a_sr=M, a_sm=1, b_sr=M, b_sq=Co * M, b_sm=1, conj_b=1,
             c_sp=Ci * M, c_sq=M, c_sm=1)
    ref = np.einsum("bom,iom->bim", g.numpy().astype(np.complex128), np.conj(w.numpy().astype(np.complex128)))
    assert rel_l2(gx.numpy(), ref) < TOL


def test_gw_contraction_conj_a(lib):
    B, Ci, Co, M = 8, 64, 64, 18
    x = _rand(B, Ci, M, seed=5)
    g = _rand(B, Co, M, seed=6)
    gw = torch.zeros(Ci, Co, M, dtype=torch.complex64)
    run_gemm(lib, x, g, gw, flags=_lib.SC_GEMM_WIDE | _lib.SC_GEMM_GRID(2), P=Ci, Q=Co, R=B, n_modes=M,
             a_sp=M, a_sr=Ci * M, a_sm=1, conj_a=1, b_sr=Co * M, b_sq=M, b_sm=1,
             c_sp=Co * M, c_sq=M, c_sm=1)
    ref = np.einsum("bim,bom->iom", np.conj(x.numpy().astype(np.complex128)), g.numpy().astype(np.complex128))
    assert rel_l2(gw.numpy(), ref) < TOL


def test_sub_block_index_tables(lib):
    """B read / C written through int32 offset tables (centred sub-block of a larger weight)."""
    B, Ci, Co, M, Wm = 32, 8, 64, 10, 23
    idx = torch.tensor([2, 3, 4, 5, 6, 12, 13, 14, 15, 16], dtype=torch.int32)
    x = _rand(B, Ci, M, seed=7)
    w = _rand(Ci, Co, Wm, seed=8)
    y = torch.zeros(B, Co, M, dtype=torch.complex64)
    run_gemm(lib, x, w, y, flags=_lib.SC_GEMM_GRID(2), P=B, Q=Co, R=Ci, n_modes=M,
             a_sp=Ci * M, a_sr=M, a_sm=1, b_sr=Co * Wm, b_sq=Wm, b_sm=1, b_idx=idx.data_ptr(),
             c_sp=Co * M, c_sq=M, c_sm=1)
    ref = np.einsum("bim,iom->bom", x.numpy().astype(np.complex128),
                    w.numpy().astype(np.complex128)[:, :, idx.numpy()])
    assert rel_l2(y.numpy(), ref) < TOL
    # scatter side: gW into the stored (larger) weight, untouched entries stay zero
    g = _rand(B, Co, M, seed=9)
    x64 = _rand(B, 64, M, seed=10)
    gw = torch.zeros(64, Co, Wm, dtype=torch.complex64)
    run_gemm(lib, x64, g, gw, flags=_lib.SC_GEMM_WIDE | _lib.SC_GEMM_GRID(2), P=64, Q=Co, R=B, n_modes=M,
             a_sp=M, a_sr=64 * M, a_sm=1, conj_a=1, b_sr=Co * M, b_sq=M, b_sm=1,
             c_sp=Co * Wm, c_sq=Wm, c_sm=1, c_idx=idx.data_ptr())
    ref = np.zeros((64, Co, Wm), dtype=np.complex128)
    ref[:, :, idx.numpy()] = np.einsum("bim,bom->iom", np.conj(x64.numpy().astype(np.complex128)),
                                       g.numpy().astype(np.complex128))
    assert rel_l2(gw.numpy(), ref) < TOL


def test_mode_summed_gemm_factor_gradient(lib):
    """sc_modegemm_msum: C[p,q] = sum_m sum_r conj(A[p,r,m]) B[r,q,m] (gradient of a Tucker factor),
    strided / transposed operands and a mode-independent B."""
    P, Q, R, M = 5, 7, 3, 70
    a = _rand(R, P, M, seed=11)            # stored [r][p][m]: transposed view of the A operand
    b = _rand(R, Q, M, seed=12)
    c = torch.zeros(P, Q, dtype=torch.complex64)
    d = _lib.ModeGemmDesc()
    for k, v in dict(P=P, Q=Q, R=R, n_modes=M, a_sp=M, a_sr=P * M, a_sm=1, conj_a=1,
                     b_sr=Q * M, b_sq=M, b_sm=1, c_sp=Q, c_sq=1).items():
        setattr(d, k, v)
    lib.modegemm_msum(torch.view_as_real(a).data_ptr(), torch.view_as_real(b).data_ptr(),
                      torch.view_as_real(c).data_ptr(), 0, **{k: getattr(d, k) for k, _ in d._fields_
                                                             if k not in ("b_idx", "c_idx")})
    ref = np.einsum("rpm,rqm->pq", np.conj(a.numpy().astype(np.complex128)), b.numpy().astype(np.complex128))
    assert rel_l2(c.numpy(), ref) < TOL
    # mode-independent B (b_sm = 0): C[p,q] = sum_m sum_r A[p,r,m] U[r,q]
    a2 = _rand(P, R, M, seed=13)
    u = _rand(R, Q, seed=14)
    c2 = torch.zeros(P, Q, dtype=torch.complex64)
    lib.modegemm_msum(torch.view_as_real(a2).data_ptr(), torch.view_as_real(u).data_ptr(),
                      torch.view_as_real(c2).data_ptr(), 0, P=P, Q=Q, R=R, n_modes=M, a_sp=R * M, a_sr=M, a_sm=1,
                      b_sr=Q, b_sq=1, b_sm=0, c_sp=Q, c_sq=1)
    ref2 = np.einsum("prm,rq->pq", a2.numpy().astype(np.complex128), u.numpy().astype(np.complex128))
    assert rel_l2(c2.numpy(), ref2) < TOL


def test_mode_independent_factor_on_matrix_cores(lib):
    """z[b,f,m] = sum_i xhat[b,i,m] U[i,f] with a mode-independent, ragged factor (b_sm = 0, 36 columns):
    the first step of the Tucker chain at TFNO-like channel counts."""
    B, Ci, F, M = 32, 64, 36, 11
    x = _rand(B, Ci, M, seed=21)
    u = _rand(Ci, F, seed=22)
    z = torch.zeros(B, F, M, dtype=torch.complex64)
    run_gemm(lib, x, u, z, flags=_lib.SC_GEMM_GRID(2), P=B, Q=F, R=Ci, n_modes=M,
             a_sp=Ci * M, a_sr=M, a_sm=1, b_sr=F, b_sq=1, b_sm=0, c_sp=F * M, c_sq=M, c_sm=1)
    ref = np.einsum("bim,if->bfm", x.numpy().astype(np.complex128), u.numpy().astype(np.complex128))
    assert rel_l2(z.numpy(), ref) < TOL


@pytest.mark.parametrize("conj_b", [0, 1])
def test_few_rows_many_modes_valu_mapping(lib, conj_b):
    """P <= 4 rows and many mode tiles on the VALU kernel (a batch of 4 at 1024^2: weight streaming): one p group,
    three of the four waves idle.  Mode count not a multiple of 64, transposed B strides as in the gX contraction."""
    P, R, Q, M = 3, 6, 10, 64 * 66 + 37
    a = _rand(P, R, M, seed=41)
    w = _rand(Q, R, M, seed=42) if conj_b else _rand(R, Q, M, seed=42)
    c = torch.full((P, Q, M), float("nan"), dtype=torch.complex64)
    kw = dict(P=P, Q=Q, R=R, n_modes=M, a_sp=R * M, a_sr=M, a_sm=1, c_sp=Q * M, c_sq=M, c_sm=1, conj_b=conj_b)
    if conj_b:
        kw.update(b_sr=M, b_sq=R * M, b_sm=1)           # B[r][q] = conj(w[q][r])
    else:
        kw.update(b_sr=Q * M, b_sq=M, b_sm=1)
    assert not lib.modegemm_uses_matrix_cores(**kw)
    lib.modegemm(_c(a).data_ptr(), _c(w).data_ptr(), torch.view_as_real(c).data_ptr(), 0, **kw)
    ref = np.einsum("prm,qrm->pqm", a.numpy().astype(np.complex128), np.conj(w.numpy().astype(np.complex128))) if conj_b \
        else np.einsum("prm,rqm->pqm", a.numpy().astype(np.complex128), w.numpy().astype(np.complex128))
    assert rel_l2(c.numpy(), ref) < TOL
