"""The oracle restatement against the committed golden vectors (made by the verbatim
reference, oracle/gen_golden.py).  CPU only."""
import numpy as np
import pytest
import torch

from conftest import DENSE_GOLDEN, FACT_GOLDEN, load_golden
from oracle import spectral_oracle as so


@pytest.mark.parametrize("name", DENSE_GOLDEN)
def test_torch_restatement_matches_golden(name):
    g = load_golden(name)
    x = torch.from_numpy(g["x"]).requires_grad_(True)
    w = torch.from_numpy(g["weight"]).requires_grad_(True)
    b = torch.from_numpy(g["bias"]).requires_grad_(True)
    torch.set_num_threads(1)
    y = so.forward_torch(x, w, b, list(g["n_modes_attr"]), list(g["max_n_modes_attr"]))
    y.backward(torch.from_numpy(g["g"]))
    # same op chain as the reference on the same machine class: fp32 round-off only
    assert so.rel_l2(y.detach().numpy(), g["y"]) < 1e-6
    assert so.rel_l2(x.grad.numpy(), g["gx"]) < 1e-6
    assert so.rel_l2(w.grad.numpy(), g["gw"]) < 1e-6
    assert so.rel_l2(b.grad.numpy(), g["gbias"]) < 1e-6


@pytest.mark.parametrize("name", DENSE_GOLDEN)
def test_np64_kept_rows_matches_golden(name):
    g = load_golden(name)
    nm, mx = list(g["n_modes_attr"]), list(g["max_n_modes_attr"])
    y, _ = so.forward_np64(g["x"], g["weight"], g["bias"], nm, mx)
    gx, gw, gb = so.backward_np64(g["x"], g["weight"], g["g"], nm, mx)
    # golden is fp32; 1e-6 is ~5x its round-off (SURVEY 8c "parity budget")
    assert so.rel_l2(y, g["y"]) < 1e-6
    assert so.rel_l2(gx, g["gx"]) < 1e-6
    assert so.rel_l2(gw, g["gw"]) < 1e-6
    assert so.rel_l2(gb, g["gbias"]) < 1e-6


@pytest.mark.parametrize("name", FACT_GOLDEN)
def test_factorized_contractions_match_golden(name):
    g = load_golden(name)
    x = torch.from_numpy(g["x"])
    nd = x.ndim - 2
    facs = [torch.from_numpy(g[f"factor_{i}"]) for i in range(nd + 2)]
    nm, mx = list(g["n_modes_attr"]), list(g["max_n_modes_attr"])
    if "core" in g:
        core = torch.from_numpy(g["core"])
        w = so.reconstruct_tucker(core, facs)
        contract = lambda xk, wk: so.contract_tucker(xk, core, facs)
    else:
        lam = torch.from_numpy(g["weights"])
        w = so.reconstruct_cp(lam, facs)
        contract = lambda xk, wk: so.contract_cp(xk, lam, facs)
    assert so.rel_l2(w.numpy(), g["w_dense"]) < 1e-6
    b = torch.from_numpy(g["bias"])
    y_dense = so.forward_torch(x, w, b, nm, mx)
    y_fact = so.forward_torch(x, w, b, nm, mx, contract=contract)
    assert so.rel_l2(y_dense.numpy(), g["y"]) < 2e-6
    assert so.rel_l2(y_fact.numpy(), g["y"]) < 2e-6


def test_weight_slices_rules():
    # even k keeps -k/2 .. k/2-1
    sl, fr = so.weight_slices([256, 256], [64, 33], [64, 33])
    assert list(fr[0][[0, -1]]) == [-32, 31] and list(fr[1][[0, -1]]) == [0, 32]
    # runtime-reduced modes take the centred sub-block
    sl, fr = so.weight_slices([16, 16], [6, 4], [8, 5])
    assert sl[0] == slice(1, -1) and sl[1] == slice(None, -1)
    assert list(fr[0]) == [-3, -2, -1, 0, 1, 2] and list(fr[1]) == [0, 1, 2, 3]
    # odd modes
    sl, fr = so.weight_slices([9, 11], [5, 4], [5, 4])
    assert list(fr[0]) == [-2, -1, 0, 1, 2]
    assert so.halve_last((64, 64)) == [64, 33]
