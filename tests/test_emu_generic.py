"""CPU tier: the engine's kernels, compiled for the host by tests/emu (same source as the
HIP build), against the golden vectors of the verbatim reference.  This validates index
maths / LDS choreography / barrier placement before any GPU time is spent; the parity
tests proper are the ``-m gpu`` ones."""
import numpy as np
import pytest
import torch

from conftest import DENSE_GOLDEN, load_golden
from engine_runner import emu_lib, layer_fwd_bwd, rel_l2
from neuraloperator_amd import _lib

TOL = 1e-5  # north-star parity bar (fp32 rel-L2)


@pytest.fixture(scope="module")
def lib():
    return emu_lib()


SMALL = [n for n in DENSE_GOLDEN if n not in ("darcy_c1_16x16_m12_c32", "d2_64x64_m16_c8", "d3_16x16x16_m8")]


@pytest.mark.parametrize("name", SMALL)
def test_generic_path_matches_golden(lib, name):
    g = load_golden(name)
    x, w, b, gy = (torch.from_numpy(g[k]) for k in ("x", "weight", "bias", "g"))
    y, gx, gw, gb, _ = layer_fwd_bwd(lib, x, w, b, gy, list(g["n_modes_attr"]), list(g["max_n_modes_attr"]),
                                     flags=_lib.SC_PLAN_FORCE_GENERIC)
    assert rel_l2(y.numpy(), g["y"]) < TOL
    assert rel_l2(gx.numpy(), g["gx"]) < TOL
    assert rel_l2(gw.numpy(), g["gw"]) < TOL
    assert rel_l2(gb.numpy(), g["gbias"]) < TOL
