"""The oracle restatement against the committed golden vectors (made by the verbatim
reference, oracle/gen_golden.py).  CPU only."""
import numpy as np
import pytest
import torch

from conftest import DENSE_GOLDEN, FACT_GOLDEN, golden_names, load_golden
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


VARIANT_GOLDEN = [n for n in golden_names() if n.startswith(("sep_", "tt_", "res_", "cplx_"))]


@pytest.mark.parametrize("name", VARIANT_GOLDEN)
def test_variant_restatement_matches_golden(name):
    """separable / TT / resolution-changing / complex-data branches of the restatement against what the
    verbatim module returned (forward and, through autograd, the input and dense-weight gradients)."""
    import json
    g = load_golden(name)
    kw = json.loads(str(g["ctor_kwargs"]))
    cplx, sep = bool(kw.get("complex_data", False)), bool(kw.get("separable", False))
    x = torch.from_numpy(g["x"]).requires_grad_(True)
    w = torch.from_numpy(g["w_dense"]).requires_grad_(True)
    b = torch.from_numpy(g["bias"]).requires_grad_(True)
    out_shape = [int(v) for v in g["y"].shape[2:]]
    torch.set_num_threads(1)
    y = so.forward_torch(x, w, b, list(g["n_modes_attr"]), list(g["max_n_modes_attr"]), separable=sep,
                         output_shape=out_shape, complex_data=cplx)
    y.backward(torch.from_numpy(g["g"]))
    assert so.rel_l2(y.detach().numpy(), g["y"]) < 2e-6
    assert so.rel_l2(x.grad.numpy(), g["gx"]) < 2e-6
    assert so.rel_l2(b.grad.numpy(), g["gbias"]) < 2e-6
    if str(g["weight_kind"]) == "DenseTensor":
        assert so.rel_l2(w.grad.numpy(), g["g_param_0"]) < 2e-6
    if str(g["weight_kind"]) == "TTTensor":
        cores = [torch.from_numpy(g[f"param_{i}"]) for i in range(x.ndim)]
        assert so.rel_l2(so.reconstruct_tt(cores).numpy(), g["w_dense"]) < 1e-6
        sl = (slice(None), slice(None)) + tuple(so.weight_slices(list(x.shape[2:]), list(g["n_modes_attr"]),
                                                                 list(g["max_n_modes_attr"]))[0])
        blk = type("W", (), {"out_channels": g["y"].shape[1]})()
        yt = so.forward_torch(x.detach(), lambda s_: blk, b.detach(), list(g["n_modes_attr"]),
                              list(g["max_n_modes_attr"]),
                              contract=lambda xk, wk: so.contract_tt(xk, [c[:, s_, :] for c, s_ in zip(cores, sl)]))
        assert so.rel_l2(yt.numpy(), g["y"]) < 2e-6


def test_mode_maps_host_logic():
    """neuraloperator_amd/modes.py: the frequency maps handed to sc_plan_create"""
    from neuraloperator_amd import modes
    assert modes.analysis_freqs([16, 16], [8, 5]) is None
    assert modes.synthesis_freqs([16, 16], [16, 16], [8, 5]) == (None, 0)
    # upsampling: negative rows stay at their INPUT-grid index (16 - 4 = 12 ...), not at the top of 32
    f, rc = modes.synthesis_freqs([16, 16], [32, 32], [8, 5])
    assert f[0] == [12, 13, 14, 15, 0, 1, 2, 3] and f[1] == [0, 1, 2, 3, 4] and rc == 0
    # downsampling: rows whose index does not exist on the small grid are dropped
    f, rc = modes.synthesis_freqs([16, 16], [8, 8], [8, 5])
    assert f[0] == [None, None, None, None, 0, 1, 2, 3] and f[1] == [0, 1, 2, 3, 4]
    # all columns kept + even output width: Im of input column n/2 is ignored
    assert modes.synthesis_freqs([8, 8], [12, 12], [8, 5])[1] == 4
    assert modes.synthesis_freqs([8, 8], [12, 11], [8, 5])[1] == 0
    # complex data: first k SHIFTED columns of the last dim; 1-d is not shifted at all
    assert modes.analysis_freqs([16, 16], [8, 6], True) == [None, [8, 9, 10, 11, 12, 13]]
    assert modes.analysis_freqs([16], [6], True) == [[0, 1, 2, 3, 4, 5]]
    assert modes.kept_block_complex([9, 11], [5, 6], [5, 6]) == ([5, 6], [0, 0])
    # resample: ceil(m/2) negative rows
    kept, fa, fs = modes.resample_block([9, 7, 6], [5, 10, 9])
    assert kept == [5, 7, 4] and fa[0] == [6, 7, 8, 0, 1] and fs[0] == [2, 3, 4, 0, 1] and fa[2] is None
