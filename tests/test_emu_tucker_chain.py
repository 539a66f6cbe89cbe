"""CPU tier: the activation side of the factorized Tucker contraction as one C-ABI call each way
(sc_tucker_chain_forward / _backward, include/sc_engine.h) in host emulation against numpy complex128 -- the pairwise
order of _contract_tucker's einsum (spectral_convolution.py:76-103) and the six products of its autograd.  Ragged ranks,
mode counts off the kernels' chunk sizes, skipped gradients, a shape that falls back to the atomic-add factor gradient."""
import numpy as np
import pytest
import torch

from engine_runner import emu_lib, rel_l2

TOL = 3e-6


@pytest.fixture(scope="module")
def lib():
    return emu_lib()


def _rand(*shape, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.complex(torch.randn(*shape, generator=g), torch.randn(*shape, generator=g))


def _p(t):
    return 0 if t is None else torch.view_as_real(t).data_ptr()


# (B, Cin, Cout, R1, R2, M)
@pytest.mark.parametrize("dims", [(3, 16, 12, 9, 7, 130), (2, 64, 64, 36, 36, 72), (4, 8, 8, 4, 5, 66), (2, 6, 5, 3, 2, 10)],
                         ids=lambda d: "B%d_Ci%d_Co%d_R%d_%d_M%d" % d)
def test_tucker_chain_matches_einsum(lib, dims):
    B, Ci, Co, R1, R2, M = dims
    xhat, u_in, t3, u_out = _rand(B, Ci, M, seed=1), _rand(Ci, R1, seed=2), _rand(R1, R2, M, seed=3), _rand(Co, R2, seed=4)
    gy = _rand(B, Co, M, seed=5)
    nan = lambda *sh: torch.full(sh, float("nan"), dtype=torch.complex64)
    z, t, yhat = nan(B, R1, M), nan(B, R2, M), nan(B, Co, M)
    lib.tucker_chain_forward(dims, _p(xhat), _p(u_in), _p(t3), _p(u_out), _p(z), _p(t), _p(yhat))
    c = lambda v: v.numpy().astype(np.complex128)
    X, Ui, T3, Uo, G = c(xhat), c(u_in), c(t3), c(u_out), c(gy)
    Z = np.einsum("bim,if->bfm", X, Ui)
    T = np.einsum("bfm,fgm->bgm", Z, T3)
    Y = np.einsum("bgm,og->bom", T, Uo)
    assert rel_l2(z.numpy(), Z) < TOL and rel_l2(t.numpy(), T) < TOL and rel_l2(yhat.numpy(), Y) < TOL
    # gradients of L = Re <gy, yhat> in torch's convention (grad = dL / d conj(param))
    gT = np.einsum("bom,og->bgm", G, np.conj(Uo))
    gUo = np.einsum("bgm,bom->og", np.conj(T), G)
    gZ = np.einsum("bgm,fgm->bfm", gT, np.conj(T3))
    gT3 = np.einsum("bfm,bgm->fgm", np.conj(Z), gT)
    gX = np.einsum("bfm,if->bim", gZ, np.conj(Ui))
    gUi = np.einsum("bim,bfm->if", np.conj(X), gZ)
    nb = lib.tucker_chain_workspace_bytes(dims)
    assert nb >= 8 * B * (R1 + R2) * M
    ws = torch.empty(nb + 64, dtype=torch.uint8)
    gx, gui, gt3, guo = nan(B, Ci, M), nan(Ci, R1), nan(R1, R2, M), nan(Co, R2)
    lib.tucker_chain_backward(dims, _p(xhat), _p(u_in), _p(t3), _p(u_out), _p(z), _p(t), _p(gy), _p(gx), _p(gui), _p(gt3),
                              _p(guo), ws.data_ptr(), nb)
    assert rel_l2(gx.numpy(), gX) < TOL and rel_l2(gt3.numpy(), gT3) < TOL
    assert rel_l2(gui.numpy(), gUi) < TOL and rel_l2(guo.numpy(), gUo) < TOL
    # skipped gradients: null pointers
    gx2 = nan(B, Ci, M)
    lib.tucker_chain_backward(dims, _p(xhat), _p(u_in), _p(t3), _p(u_out), _p(z), _p(t), _p(gy), _p(gx2), 0, 0, 0,
                              ws.data_ptr(), nb)
    assert torch.equal(torch.view_as_real(gx2), torch.view_as_real(gx))
    with pytest.raises(RuntimeError):
        lib.tucker_chain_backward(dims, _p(xhat), _p(u_in), _p(t3), _p(u_out), _p(z), _p(t), _p(gy), _p(gx2), 0, 0, 0,
                                  ws.data_ptr(), 16)


# ---- round 5: the same products as ONE launch each way (sc_kernels_tkchain.h) --------------------------------------
# (B, Cin, Cout, R1, R2, M): one and two row tiles of the batch, ranks below / above one column tile, fewer tiles than a
# workgroup's modes would need a second round for
@pytest.mark.parametrize("dims", [(4, 8, 8, 4, 8, 16), (8, 16, 12, 12, 20, 40), (32, 64, 64, 36, 36, 4), (20, 24, 32, 36, 28, 8)],
                         ids=lambda d: "B%d_Ci%d_Co%d_R%d_%d_M%d" % d)
def test_fused_chain_matches_einsum_and_the_nine_launches(lib, dims, monkeypatch):
    B, Ci, Co, R1, R2, M = dims
    assert not lib.tucker_chain_fused_supported(dims)          # opt-in: the nine launches are the (faster) default
    monkeypatch.setenv("SC_TKC", "1")
    assert lib.tucker_chain_fused_supported(dims)
    xhat, u_in, t3, u_out = _rand(B, Ci, M, seed=1), _rand(Ci, R1, seed=2), _rand(R1, R2, M, seed=3), _rand(Co, R2, seed=4)
    gy = _rand(B, Co, M, seed=5)
    nan = lambda *sh: torch.full(sh, float("nan"), dtype=torch.complex64)
    z, t, yhat, t3m = nan(B, R1, M), nan(B, R2, M), nan(B, Co, M), nan(M, R1, R2)
    assert lib.tucker_chain_t3m_bytes(dims) == 8 * M * R1 * R2
    lib.tucker_chain_forward_fused(dims, _p(xhat), _p(u_in), _p(t3), _p(u_out), _p(t3m), _p(z), _p(t), _p(yhat))
    assert torch.equal(torch.view_as_real(t3m), torch.view_as_real(t3.permute(2, 0, 1).contiguous()))
    c = lambda v: v.numpy().astype(np.complex128)
    X, Ui, T3, Uo, G = c(xhat), c(u_in), c(t3), c(u_out), c(gy)
    Z = np.einsum("bim,if->bfm", X, Ui)
    T = np.einsum("bfm,fgm->bgm", Z, T3)
    Y = np.einsum("bgm,og->bom", T, Uo)
    assert rel_l2(z.numpy(), Z) < TOL and rel_l2(t.numpy(), T) < TOL and rel_l2(yhat.numpy(), Y) < TOL
    # ... and next to the three launches it replaces (other kernels, other summation orders: fp32 round-off apart)
    z9, t9, y9 = nan(B, R1, M), nan(B, R2, M), nan(B, Co, M)
    lib.tucker_chain_forward(dims, _p(xhat), _p(u_in), _p(t3), _p(u_out), _p(z9), _p(t9), _p(y9))
    assert rel_l2(z.numpy(), z9.numpy()) < 1e-6 and rel_l2(t.numpy(), t9.numpy()) < 1e-6 and rel_l2(yhat.numpy(), y9.numpy()) < 1e-6
    gT = np.einsum("bom,og->bgm", G, np.conj(Uo))
    gUo = np.einsum("bgm,bom->og", np.conj(T), G)
    gZ = np.einsum("bgm,fgm->bfm", gT, np.conj(T3))
    gT3 = np.einsum("bfm,bgm->fgm", np.conj(Z), gT)
    gX = np.einsum("bfm,if->bim", gZ, np.conj(Ui))
    gUi = np.einsum("bim,bfm->if", np.conj(X), gZ)
    nb = lib.tucker_chain_backward_fused_workspace_bytes(dims)
    assert nb >= 8 * M * R1 * R2
    ws = torch.empty(nb + 64, dtype=torch.uint8)
    gx, gui, gt3, guo = nan(B, Ci, M), nan(Ci, R1), nan(R1, R2, M), nan(Co, R2)
    lib.tucker_chain_backward_fused(dims, _p(xhat), _p(u_in), _p(t3m), _p(u_out), _p(z), _p(t), _p(gy), _p(gx), _p(gui),
                                    _p(gt3), _p(guo), ws.data_ptr(), nb)
    assert rel_l2(gx.numpy(), gX) < TOL and rel_l2(gt3.numpy(), gT3) < TOL
    assert rel_l2(gui.numpy(), gUi) < TOL and rel_l2(guo.numpy(), gUo) < TOL
    # run-to-run identical (fixed-order reduction), also with the input gradient skipped
    gui2, gt32, guo2 = nan(Ci, R1), nan(R1, R2, M), nan(Co, R2)
    lib.tucker_chain_backward_fused(dims, _p(xhat), _p(u_in), _p(t3m), _p(u_out), _p(z), _p(t), _p(gy), 0, _p(gui2),
                                    _p(gt32), _p(guo2), ws.data_ptr(), nb)
    for a, b in ((gui, gui2), (gt3, gt32), (guo, guo2)):
        assert torch.equal(torch.view_as_real(a), torch.view_as_real(b))
    with pytest.raises(RuntimeError):
        lib.tucker_chain_backward_fused(dims, _p(xhat), _p(u_in), _p(t3m), _p(u_out), _p(z), _p(t), _p(gy), 0, _p(gui2),
                                        _p(gt32), _p(guo2), ws.data_ptr(), 16)


@pytest.mark.parametrize("dims", [(3, 16, 12, 9, 7, 130), (64, 64, 64, 36, 36, 8), (4, 8, 8, 4, 8, 18), (4, 128, 8, 4, 8, 16)],
                         ids=lambda d: "B%d_Ci%d_Co%d_R%d_%d_M%d" % d)
def test_fused_chain_refuses_shapes_outside_its_limits(lib, dims, monkeypatch):
    monkeypatch.setenv("SC_TKC", "1")
    assert not lib.tucker_chain_fused_supported(dims)
