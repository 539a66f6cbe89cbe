"""CPU tier: T[f, g, x, y] = sum_{c, d} core[f, g, c, d] U_x[x, c] U_y[y, d] in one launch each way
(sc_kernels_tucker.h; the batch-independent part of _contract_tucker, spectral_convolution.py:76-103) in host emulation
against torch's complex128 einsum and its autograd; the sizes of BASELINE configs[2] (ranks (36, 36, 36, 19), kept
64 x 33) at a reduced number of (f, g) pairs, ragged small sizes, the size limits."""
import pytest
import torch

from engine_runner import emu_lib, rel_l2


@pytest.fixture(scope="module")
def lib():
    return emu_lib()


def _c(*shape, g):
    return torch.complex(torch.randn(*shape, generator=g), torch.randn(*shape, generator=g))


@pytest.mark.parametrize("fg,rx,ry,mx,my", [(40, 36, 19, 64, 33), (7, 3, 5, 6, 4), (600, 4, 2, 8, 5), (3, 64, 16, 64, 64)])
def test_tucker_mode_factors(lib, fg, rx, ry, mx, my):
    g = torch.Generator().manual_seed(fg)
    core, ux, uy, gt = _c(fg, rx, ry, g=g), _c(mx, rx, g=g), _c(my, ry, g=g), _c(fg, mx, my, g=g)
    assert lib.tucker_modes_supported(fg, rx, ry, mx, my)
    t = torch.full((fg, mx, my), float("nan"), dtype=torch.complex64)
    p = lambda z: torch.view_as_real(z).data_ptr()
    lib.tucker_modes_forward(fg, rx, ry, mx, my, p(core), p(ux), p(uy), p(t), 0)
    cd, ad, bd = (z.to(torch.complex128).requires_grad_(True) for z in (core, ux, uy))
    ref = torch.einsum("ncd,xc,yd->nxy", cd, ad, bd)
    assert rel_l2(t.numpy(), ref.detach().numpy()) < 2e-6
    ref.backward(gt.to(torch.complex128))
    gc, ga, gb = torch.empty_like(core), torch.empty_like(ux), torch.empty_like(uy)
    ws = torch.empty(lib.tucker_modes_workspace_bytes(fg, rx, ry, mx, my), dtype=torch.uint8)
    lib.tucker_modes_backward(fg, rx, ry, mx, my, p(core), p(ux), p(uy), p(gt), p(gc), p(ga), p(gb), ws.data_ptr(), 0)
    assert rel_l2(gc.numpy(), cd.grad.numpy()) < 5e-6
    assert rel_l2(ga.numpy(), ad.grad.numpy()) < 5e-6
    assert rel_l2(gb.numpy(), bd.grad.numpy()) < 5e-6


def test_limits(lib):
    assert lib.tucker_modes_supported(10, 64, 16, 64, 64)
    assert not lib.tucker_modes_supported(10, 36, 19, 128, 33)      # more than 64 modes along a dim
    assert not lib.tucker_modes_supported(10, 36, 40, 64, 33)       # My Ry beyond a thread set
    assert lib.tucker_modes_workspace_bytes(10, 36, 19, 128, 33) == 0
