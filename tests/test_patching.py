"""Multigrid patching against the verbatim reference class on the CPU:
neuraloperator_amd.mpu.patching.MultigridPatching2D / make_patches == neuralop/training/patching.py for every
configuration the reference's own tests use (levels, padding fractions, stitching) and a few more."""
import io
import contextlib

import pytest
import torch

from neuraloperator_amd.mpu import patching as ours
from oracle import ref_verbatim

pytestmark = pytest.mark.skipif(not ref_verbatim.available(), reason="verbatim reference not present")


def _quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)


@pytest.mark.parametrize("n,p", [(2, 0), (4, 1), ([2, 4], [3, 0]), (1, 2), (1, 0)])
@pytest.mark.parametrize("shape", [(2, 3, 16, 16), (1, 2, 8, 24)])
def test_make_patches(shape, n, p):
    ref = ref_verbatim.load_reference_patching()
    x = torch.randn(*shape)
    assert torch.equal(ours.make_patches(x, n, p), ref.make_patches(x, n, p))


def test_make_patches_1d():
    """(B, C, S) inputs: the reference's docstring promises them but its code raises (a 6-dim permute of a 4-dim
    tensor, patching.py:367); here they work -- checked against plain slicing of the periodic signal."""
    x = torch.randn(2, 3, 32)
    got = ours.make_patches(x, 4, 2)
    assert got.shape == (8, 3, 12)
    xp = torch.cat([x[..., -2:], x, x[..., :2]], dim=-1)
    for b in range(2):
        for i in range(4):
            assert torch.equal(got[b * 4 + i], xp[b, :, 8 * i:8 * i + 12])


@pytest.mark.parametrize("levels", [0, 1, 2, 3])
@pytest.mark.parametrize("padding", [0, 0.1, 0.25, [0.125, 0.0]])
@pytest.mark.parametrize("stitching", [True, False])
@pytest.mark.parametrize("size", [(32, 32), (64, 32)])
def test_patch_unpatch(levels, padding, stitching, size):
    ref = ref_verbatim.load_reference_patching()
    model = torch.nn.Conv2d(1, 1, 1)
    a = _quiet(ours.MultigridPatching2D, model, levels=levels, padding_fraction=padding, stitching=stitching)
    b = _quiet(ref.MultigridPatching2D, model, levels=levels, padding_fraction=padding, stitching=stitching)
    x = torch.randn(3, 2, *size)
    y = torch.randn(3, 1, *size)
    try:
        xa, ya = a.patch(x, y)
    except ValueError:
        # more coarse windows than patches (e.g. 32 x 32 with levels = 3): the reference's reshape cannot fold them
        # either (patching.py:272-281 raises a RuntimeError); here the condition is named
        with pytest.raises(RuntimeError):
            b.patch(x, y)
        return
    xb, yb = b.patch(x, y)
    assert xa.shape == xb.shape and torch.equal(xa, xb) and torch.equal(ya, yb)
    assert (a.padding_height, a.padding_width) == (getattr(b, "padding_height", 0), getattr(b, "padding_width", 0))
    # the model output has the patch layout of x with the model's channel count: unpatch it both ways
    out = torch.randn(xa.shape[0], 1, *xa.shape[2:])
    for evaluation in (False, True):
        ua, va = a.unpatch(out, ya, evaluation=evaluation)
        if levels == 0:
            # the reference never sets padding_height / padding_width without levels and raises in unpatch
            # (patching.py:128, 208-210); here no patching is the identity
            with pytest.raises(AttributeError):
                b.unpatch(out, yb, evaluation=evaluation)
            assert torch.equal(ua, out) and torch.equal(va, ya)
            continue
        ub, vb = b.unpatch(out, yb, evaluation=evaluation)
        if (a.padding_height == 0) != (a.padding_width == 0):
            # a halo in one dim only: the reference slices [0:-0] in the other and returns an EMPTY tensor
            # (patching.py:297-301); here the un-haloed dim is left alone
            assert ub.numel() == 0 and ua.numel() > 0
            continue
        assert torch.equal(ua, ub) and torch.equal(va, vb)
    if stitching and levels > 0:
        # the fine channels of the patches stitch back to the input
        fine = a._unpad(xa[:, :x.shape[1]]) if (a.padding_height or a.padding_width) else xa[:, :x.shape[1]]
        assert torch.equal(a._stitch(fine), x)


def test_halo_wider_than_the_coarse_field():
    """levels 2 with a 29-pixel halo on a 64 x 64 grid: the coarse views are padded by more than their own size (the reference pads twice
    there, patching.py:247-270); one periodic gather covers it."""
    ref = ref_verbatim.load_reference_patching()
    model = torch.nn.Identity()
    a = _quiet(ours.MultigridPatching2D, model, levels=2, padding_fraction=0.45)
    b = _quiet(ref.MultigridPatching2D, model, levels=2, padding_fraction=0.45)
    x = torch.randn(2, 3, 64, 64)
    assert torch.equal(a._make_mg_patches(x), b._make_mg_patches(x))
