"""CPU tier: the small-extent streaming contraction (sc_kernels_sb.h, k_modegemm_sb: a lane owns two neighbouring
modes, PT x QT register tile, a hand-written ring of reduction steps in flight) in host emulation against a numpy
complex128 einsum and, bit for bit, against the lanes-are-modes VALU kernel it replaces for these shapes.  Covers the
three contractions of a layer at a small batch (BASELINE configs[4]: B = 4) -- forward, gX with conj(B) through
transposed strides, gW with conj(A) and a reduction of B terms -- mode counts that do not fill the 512-mode tile,
ragged row / column tiles, reductions shorter and longer than the ring, and the dispatch rule."""
import numpy as np
import pytest
import torch

from engine_runner import emu_lib, rel_l2
from neuraloperator_amd import _lib

TOL = 1e-5


@pytest.fixture(scope="module")
def lib():
    return emu_lib()


def _rand(*shape, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.complex(torch.randn(*shape, generator=g), torch.randn(*shape, generator=g))


def _run(lib, a, b, c, flags=0, expect=3, **kw):
    assert lib.modegemm_path(flags=flags, **kw) == expect, "test must exercise the intended kernel"
    lib.modegemm(torch.view_as_real(a).data_ptr(), torch.view_as_real(b).data_ptr(),
                 torch.view_as_real(c).data_ptr(), 0, flags=flags, **kw)
    return c


def _both(lib, a, b, shape, ref, **kw):
    c = torch.full(shape, float("nan"), dtype=torch.complex64)
    _run(lib, a, b, c, **kw)
    assert rel_l2(c.numpy(), ref) < TOL
    c4 = torch.full(shape, float("nan"), dtype=torch.complex64)           # the other wave arrangement: same bits
    _run(lib, a, b, c4, flags=_lib.SC_GEMM_SB_WM4, **kw)
    assert torch.equal(torch.view_as_real(c), torch.view_as_real(c4))
    c0 = torch.full(shape, float("nan"), dtype=torch.complex64)           # the kernel it replaces: same fmaf chains
    fl = _lib.SC_GEMM_NO_SB | _lib.SC_GEMM_FORCE_VALU | _lib.SC_GEMM_NO_STREAM
    _run(lib, a, b, c0, flags=fl, expect=0, **kw)
    assert torch.equal(torch.view_as_real(c), torch.view_as_real(c0))


# (B, Ci, Co, M)
@pytest.mark.parametrize("dims", [(4, 9, 12, 70), (3, 5, 6, 2), (4, 128, 8, 516), (1, 7, 5, 1026), (2, 2, 3, 130)],
                         ids=lambda d: "B%d_Ci%d_Co%d_M%d" % d)
def test_forward_small_batch(lib, dims):
    B, Ci, Co, M = dims
    x, w = _rand(B, Ci, M, seed=1), _rand(Ci, Co, M, seed=2)
    ref = np.einsum("bim,iom->bom", x.numpy().astype(np.complex128), w.numpy().astype(np.complex128))
    _both(lib, x, w, (B, Co, M), ref, P=B, Q=Co, R=Ci, n_modes=M, a_sp=Ci * M, a_sr=M, a_sm=1, b_sr=Co * M, b_sq=M,
          b_sm=1, c_sp=Co * M, c_sq=M, c_sm=1)


@pytest.mark.parametrize("dims", [(4, 10, 7, 66), (2, 6, 130, 8), (4, 5, 3, 600)], ids=lambda d: "B%d_Ci%d_Co%d_M%d" % d)
def test_gx_conj_b_transposed(lib, dims):
    B, Ci, Co, M = dims
    g, w = _rand(B, Co, M, seed=3), _rand(Ci, Co, M, seed=4)
    ref = np.einsum("bom,iom->bim", g.numpy().astype(np.complex128), np.conj(w.numpy().astype(np.complex128)))
    _both(lib, g, w, (B, Ci, M), ref, P=B, Q=Ci, R=Co, n_modes=M, a_sp=Co * M, a_sr=M, a_sm=1, b_sr=M, b_sq=Co * M,
          b_sm=1, conj_b=1, c_sp=Ci * M, c_sq=M, c_sm=1)


@pytest.mark.parametrize("dims", [(4, 10, 7, 66), (3, 128, 5, 20), (1, 6, 9, 514), (2, 33, 34, 4)],
                         ids=lambda d: "B%d_Ci%d_Co%d_M%d" % d)
@pytest.mark.parametrize("stream", [0, 1])
def test_gw_conj_a_short_reduction(lib, dims, stream):
    B, Ci, Co, M = dims
    x, g = _rand(B, Ci, M, seed=5), _rand(B, Co, M, seed=6)
    ref = np.einsum("bim,bom->iom", np.conj(x.numpy().astype(np.complex128)), g.numpy().astype(np.complex128))
    gw = torch.full((Ci, Co, M), float("nan"), dtype=torch.complex64)
    kw = dict(P=Ci, Q=Co, R=B, n_modes=M, a_sp=M, a_sr=Ci * M, a_sm=1, conj_a=1, b_sr=Co * M, b_sq=M, b_sm=1,
              c_sp=Co * M, c_sq=M, c_sm=1)
    _run(lib, x, g, gw, flags=_lib.SC_GEMM_STREAM_C if stream else 0, **kw)
    assert rel_l2(gw.numpy(), ref) < TOL
    gw4 = torch.full((Ci, Co, M), float("nan"), dtype=torch.complex64)
    _run(lib, x, g, gw4, flags=_lib.SC_GEMM_SB_WM4, **kw)
    assert torch.equal(torch.view_as_real(gw), torch.view_as_real(gw4))


def test_dispatch_rule(lib):
    """Taken for min(P, R) <= 4 on plain, even, aligned operands -- and never otherwise."""
    base = dict(P=4, Q=16, R=16, n_modes=64, a_sp=16 * 64, a_sr=64, a_sm=1, b_sr=16 * 64, b_sq=64, b_sm=1,
                c_sp=16 * 64, c_sq=64, c_sm=1)
    assert lib.modegemm_path(**base) == 3
    assert not lib.modegemm_uses_matrix_cores(**base)
    assert lib.modegemm_path(**dict(base, P=5)) != 3                      # 5 rows: the older kernels keep it
    assert lib.modegemm_path(**dict(base, P=32, R=4)) == 3                # short reduction
    assert lib.modegemm_path(**dict(base, n_modes=63)) != 3               # odd mode count
    assert lib.modegemm_path(**dict(base, a_sr=65)) != 3                  # a row that is not 16-byte aligned
    assert lib.modegemm_path(**dict(base, accumulate=1)) != 3
    assert lib.modegemm_path(flags=_lib.SC_GEMM_NO_SB, **base) != 3
    assert lib.modegemm_path(flags=_lib.SC_GEMM_F16, **base) != 3


def test_layer_at_small_batch_takes_it(lib):
    """sc_layer_forward / _backward at B = 4 (the dense layer's three contractions) against the oracle."""
    from engine_runner import layer_fwd_bwd
    from oracle import spectral_oracle as so
    torch.manual_seed(9)
    b, ci, co, spatial, modes = 4, 6, 5, (16, 12), (8, 6)
    nm = so.halve_last(modes)
    x = torch.randn(b, ci, *spatial)
    w = torch.randn(ci, co, *nm, dtype=torch.cfloat) * 0.4
    bias = torch.randn(co, 1, 1)
    g = torch.randn(b, co, *spatial)
    xc, wc, bc = x.clone().requires_grad_(True), w.clone().requires_grad_(True), bias.clone().requires_grad_(True)
    yo = so.forward_torch(xc, wc, bc, nm, nm)
    yo.backward(g)
    y, gx, gw, gb, _ = layer_fwd_bwd(lib, x, w, bias, g, nm, nm)
    assert rel_l2(y.numpy(), yo.detach().numpy()) < TOL
    assert rel_l2(gx.numpy(), xc.grad.numpy()) < TOL
    assert rel_l2(gw.numpy(), wc.grad.numpy()) < TOL
    assert rel_l2(gb.numpy(), bc.grad.numpy()) < TOL


# ---- k_modegemm_bfac: a mode-independent right operand (factor matrix) read through the scalar cache -----------------
@pytest.mark.parametrize("dims", [(3, 64, 36, 130), (2, 36, 64, 70), (4, 10, 19, 64), (1, 5, 8, 200), (2, 7, 40, 66)],
                         ids=lambda d: "P%d_R%d_Q%d_M%d" % d)
@pytest.mark.parametrize("transposed", [False, True], ids=["B_rq", "B_qr"])
@pytest.mark.parametrize("conj", [(0, 0), (0, 1), (1, 0)], ids=["plain", "conjB", "conjA"])
def test_factor_operand(lib, dims, transposed, conj):
    """C[p,q,m] = sum_r opA(A[p,r,m]) opB(B[r,q]) with B stored [r][q] or [q][r] (z = xhat U_in, yhat = t U_out^T and
    the adjoints of the Tucker / CP chains): against complex128 and, bit for bit, the VALU kernel."""
    P, R, Q, M = dims
    ca, cb = conj
    a = _rand(P, R, M, seed=7)
    bm = _rand(R, Q, seed=8)
    store = bm.t().contiguous() if transposed else bm.contiguous()
    kw = dict(P=P, Q=Q, R=R, n_modes=M, a_sp=R * M, a_sr=M, a_sm=1, b_sm=0, conj_a=ca, conj_b=cb,
              b_sr=(1 if transposed else Q), b_sq=(R if transposed else 1), c_sp=Q * M, c_sq=M, c_sm=1)
    a128, b128 = a.numpy().astype(np.complex128), bm.numpy().astype(np.complex128)
    ref = np.einsum("prm,rq->pqm", np.conj(a128) if ca else a128, np.conj(b128) if cb else b128)
    c = torch.full((P, Q, M), float("nan"), dtype=torch.complex64)
    _run(lib, a, store, c, flags=_lib.SC_GEMM_NO_FMX, expect=4, **kw)
    assert rel_l2(c.numpy(), ref) < TOL
    c0 = torch.full((P, Q, M), float("nan"), dtype=torch.complex64)
    fl = _lib.SC_GEMM_NO_SB | _lib.SC_GEMM_FORCE_VALU | _lib.SC_GEMM_NO_STREAM
    _run(lib, a, store, c0, flags=fl, expect=0, **kw)
    assert torch.equal(torch.view_as_real(c), torch.view_as_real(c0))


# ---- the two contractions of a small-batch backward pass in ONE pass over the weight (k_modegemm_sb_bwd, session 2)
# (B, Ci, Co, M): ragged row tiles (Ci % 4, < 16), mode counts off the 128-mode tile, reductions shorter / longer than
# the ring, every batch size the kernel is instantiated for
@pytest.mark.parametrize("dims", [(4, 16, 16, 128), (4, 9, 12, 70), (3, 5, 17, 2), (1, 18, 5, 258), (2, 33, 3, 130),
                                  (4, 128, 2, 20)], ids=lambda d: "B%d_Ci%d_Co%d_M%d" % d)
def test_backward_pair_one_pass_over_the_weight(lib, dims):
    B, Ci, Co, M = dims
    xh, gh, w = _rand(B, Ci, M, seed=1), _rand(B, Co, M, seed=2), _rand(Ci, Co, M, seed=3)
    kw_w = dict(P=Ci, Q=Co, R=B, n_modes=M, a_sp=M, a_sr=Ci * M, a_sm=1, conj_a=1, b_sr=Co * M, b_sq=M, b_sm=1,
                c_sp=Co * M, c_sq=M, c_sm=1, flags=_lib.SC_GEMM_STREAM_C)
    kw_x = dict(P=B, Q=Ci, R=Co, n_modes=M, a_sp=Co * M, a_sr=M, a_sm=1, b_sr=M, b_sq=Co * M, b_sm=1, conj_b=1,
                c_sp=Ci * M, c_sq=M, c_sm=1)
    p = lambda t: torch.view_as_real(t).data_ptr()
    nan = lambda *sh: torch.full(sh, float("nan"), dtype=torch.complex64)
    gw, gx = nan(Ci, Co, M), nan(B, Ci, M)
    lib.modegemm_pair(kw_w, p(xh), p(gh), p(gw), kw_x, p(gh), p(w), p(gx))
    c = lambda v: v.numpy().astype(np.complex128)
    assert rel_l2(gw.numpy(), np.einsum("bim,bom->iom", np.conj(c(xh)), c(gh))) < TOL
    assert rel_l2(gx.numpy(), np.einsum("bom,iom->bim", c(gh), np.conj(c(w)))) < TOL
    # the two launches it replaces: same fmaf chains, same bits
    gw2, gx2 = nan(Ci, Co, M), nan(B, Ci, M)
    lib.modegemm(p(xh), p(gh), p(gw2), 0, **kw_w)
    lib.modegemm(p(gh), p(w), p(gx2), 0, **kw_x)
    if Ci * Co >= 64 * B:                       # the one-pass kernel ran (else the pair is those two launches anyway)
        assert lib.modegemm_path(**kw_w) == 3 and lib.modegemm_path(**kw_x) == 3
    assert torch.equal(torch.view_as_real(gw), torch.view_as_real(gw2))
    assert torch.equal(torch.view_as_real(gx), torch.view_as_real(gx2))
    # round 5: the pair walks its mode tiles FASTEST by default (neighbouring workgroups stream neighbouring pieces of the
    # same weight rows), the single launches slowest; SC_GEMM_SB_ALT_ORDER flips either -- a different assignment of the
    # same work items: same bits
    alt = _lib.SC_GEMM_SB_ALT_ORDER
    gw3, gx3 = nan(Ci, Co, M), nan(B, Ci, M)
    lib.modegemm_pair(dict(kw_w, flags=kw_w["flags"] | alt), p(xh), p(gh), p(gw3), kw_x, p(gh), p(w), p(gx3))
    assert torch.equal(torch.view_as_real(gw), torch.view_as_real(gw3))
    assert torch.equal(torch.view_as_real(gx), torch.view_as_real(gx3))
    gw4, gx4 = nan(Ci, Co, M), nan(B, Ci, M)
    lib.modegemm(p(xh), p(gh), p(gw4), 0, **dict(kw_w, flags=kw_w["flags"] | alt))
    lib.modegemm(p(gh), p(w), p(gx4), 0, **dict(kw_x, flags=alt))
    assert torch.equal(torch.view_as_real(gw), torch.view_as_real(gw4))
    assert torch.equal(torch.view_as_real(gx), torch.view_as_real(gx4))
