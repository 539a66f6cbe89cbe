"""CPU tier: the fused AdamW step (sc_adamw_step) in host emulation against parameter / state trajectories
of the verbatim reference optimizer (neuralop/training/adamw.py, golden vectors from oracle/gen_golden.py)."""
import json

import numpy as np
import pytest
import torch

from conftest import golden_names, load_golden
from engine_runner import emu_lib, rel_l2

TOL = 2e-6


@pytest.fixture(scope="module")
def lib():
    return emu_lib()


@pytest.mark.parametrize("name", golden_names("adamw_"))
def test_fused_adamw_matches_reference_trajectory(lib, name):
    g = load_golden(name)
    kw = json.loads(str(g["kwargs"]))
    lr, (b1, b2) = kw.get("lr", 1e-3), kw.get("betas", (0.9, 0.999))
    eps, wd, cb = kw.get("eps", 1e-6), kw.get("weight_decay", 0.0), kw.get("correct_bias", True)
    for tag, cplx in (("c", True), ("r", False)):
        p = torch.from_numpy(g[f"p{tag}0"]).clone().contiguous()
        m, v = torch.zeros_like(p), torch.zeros_like(p)
        view = torch.view_as_real if cplx else (lambda t: t)
        for t in range(int(g["steps"])):
            gr = torch.from_numpy(g[f"g{tag}_{t}"]).contiguous()
            lib.adamw_step(view(p).data_ptr(), view(gr).data_ptr(), view(m).data_ptr(), view(v).data_ptr(),
                           p.numel(), cplx, 0, lr=lr, beta1=b1, beta2=b2, eps=eps, weight_decay=wd,
                           correct_bias=cb, step=t + 1)
            if t == 0:
                assert rel_l2(p.numpy(), g[f"p{tag}_after1"]) < TOL
        assert rel_l2(p.numpy(), g[f"p{tag}"]) < TOL
        assert rel_l2(m.numpy(), g[f"m_{tag}"]) < TOL
        assert rel_l2(v.numpy(), g[f"v_{tag}"]) < TOL
        if cplx:
            assert np.all(v.numpy().imag == 0)            # the reference's complex exp_avg_sq stays real-valued


def test_adamw_argument_checks(lib):
    p = torch.zeros(4)
    with pytest.raises(Exception):
        lib.adamw_step(p.data_ptr(), p.data_ptr(), p.data_ptr(), p.data_ptr(), 4, False, 0, lr=1e-3, beta1=0.9,
                       beta2=0.999, eps=1e-6, weight_decay=0.0, correct_bias=True, step=0)   # steps count from 1
    lib.adamw_step(0, 0, 0, 0, 0, False, 0, lr=1e-3, beta1=0.9, beta2=0.999, eps=1e-6, weight_decay=0.0,
                   correct_bias=True, step=1)                                                 # empty: no-op


@pytest.mark.parametrize("name", golden_names("adamw_"))
def test_adamw_class_elementwise_route_matches_reference_trajectory(name):
    """neuraloperator_amd.AdamW on parameters the fused launch does not take (CPU tensors here; bf16 / fp64 /
    non-contiguous parameters on the GPU take the same branch): same trajectory, same state layout as the
    verbatim optimizer (adamw.py:155-200)."""
    from neuraloperator_amd import AdamW
    g = load_golden(name)
    kw = json.loads(str(g["kwargs"]))
    pc = torch.nn.Parameter(torch.from_numpy(g["pc0"]).clone())
    pr = torch.nn.Parameter(torch.from_numpy(g["pr0"]).clone())
    opt = AdamW([pc, pr], **kw)
    for t in range(int(g["steps"])):
        pc.grad = torch.from_numpy(g[f"gc_{t}"]).clone()
        pr.grad = torch.from_numpy(g[f"gr_{t}"]).clone()
        opt.step()
        if t == 0:
            assert rel_l2(pc.detach().numpy(), g["pc_after1"]) < TOL
    assert rel_l2(pc.detach().numpy(), g["pc"]) < TOL and rel_l2(pr.detach().numpy(), g["pr"]) < TOL
    st = opt.state[pc]
    assert st["step"] == int(g["steps"]) and st["exp_avg"].dtype == torch.complex64
    assert rel_l2(st["exp_avg"].numpy(), g["m_c"]) < TOL and rel_l2(st["exp_avg_sq"].numpy(), g["v_c"]) < TOL
    assert rel_l2(opt.state[pr]["exp_avg_sq"].numpy(), g["v_r"]) < TOL
    sd = opt.state_dict()                                   # round trip through the reference's state-dict layout
    opt2 = AdamW([pc, pr], **kw)
    opt2.load_state_dict(sd)
    assert opt2.state[pc]["step"] == st["step"]
