"""CPU tier: the spatially decomposed layer's round-6 variants (runtime n_modes, a grid smaller than the modes,
complex_data, a change of resolution along every dim) with the ENGINE's stage ops on the host-emulation build, one rank,
against the CPU oracle (spectral_convolution.py:400-415, 465-559).  The sharding itself:
tests/test_spatial_parallel_gloo.py (world 2); the same cases on the device: tests/test_gpu_parity.py."""
import os
import sys

import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))

from oracle import spectral_oracle as so           # noqa: E402  (test infrastructure)

TOL = 1e-5


def _num(t):
    return torch.view_as_real(t.detach().contiguous()).numpy() if t.is_complex() else t.detach().numpy()


@pytest.mark.parametrize("cfg", [
    dict(spatial=(16, 12), modes=(8, 6), run_modes=(6, 4)),
    dict(spatial=(8, 8, 6), modes=(6, 5, 4), run_modes=(4, 3, 4)),
    dict(spatial=(8, 8, 6), modes=(6, 6, 6), run_modes=(4, 2, 2), fac="tucker"),
    dict(spatial=(4, 6), modes=(8, 6)),
    dict(spatial=(16, 12), modes=(8, 6), complex=True),
    dict(spatial=(8, 6, 6), modes=(4, 4, 3), complex=True, run_modes=(3, 4, 2)),
    dict(spatial=(16, 12), modes=(8, 6), complex=True, out_shape=(24, 10)),
    dict(spatial=(8, 8, 6), modes=(4, 4, 4), out_shape=(8, 12, 6)),
    dict(spatial=(8, 8, 6), modes=(6, 6, 4), out_shape=(12, 5, 10)),
    dict(spatial=(16, 12), modes=(8, 6), separable=True, run_modes=(6, 4)),
], ids=lambda c: "-".join(f"{k}={v}" for k, v in c.items()).replace(" ", ""))
def test_pencil_layer_variants_on_the_emulated_engine(cfg):
    from emu_engine import engine_on_emulation
    from neuraloperator_amd.modes import halve_last_mode
    from neuraloperator_amd.mpu import SpatialParallelSpectralConv

    torch.manual_seed(31)
    spatial, modes, cplx, fac = cfg["spatial"], cfg["modes"], cfg.get("complex", False), cfg.get("fac", "dense")
    out_shape, sep = cfg.get("out_shape"), cfg.get("separable", False)
    co = 4 if sep else 3
    mx = halve_last_mode(modes, cplx)
    dt = torch.cfloat if cplx else torch.float32
    with engine_on_emulation():
        sp = SpatialParallelSpectralConv(4, co, modes, factorization=fac, rank=0.5, complex_data=cplx, separable=sep)
        assert sp.P == 1
        if cfg.get("run_modes") is not None:
            sp.n_modes = cfg["run_modes"]
        nm = list(sp.n_modes)
        x = torch.randn(2, 4, *spatial, dtype=dt, requires_grad=True)
        og = list(out_shape) if out_shape is not None else list(spatial)
        g = torch.randn(2, co, *og, dtype=dt)
        y = sp(x, output_shape=out_shape)
        assert list(y.shape) == [2, co, *og]
        y.backward(g)
    xc = x.detach().clone().requires_grad_(True)
    bc = sp.bias.detach().clone().requires_grad_(True)
    if fac == "dense":
        wc = sp.weight.detach().clone().requires_grad_(True)
    else:
        from neuraloperator_amd.factorized import SpectralWeight
        ref = SpectralWeight.new(((4,) if sep else (4, 3)) + tuple(mx), rank=0.5, factorization=fac)
        with torch.no_grad():
            for q, r in zip(ref.parameters(), sp.weight.parameters()):
                q.copy_(r)
        wc = ref.to_tensor()
    yo = so.forward_torch(xc, wc, bc, nm, mx, output_shape=out_shape, complex_data=cplx, separable=sep)
    yo.backward(g)
    assert so.rel_l2(_num(y), _num(yo)) < TOL
    assert so.rel_l2(_num(x.grad), _num(xc.grad)) < TOL
    assert so.rel_l2(_num(sp.bias.grad), _num(bc.grad)) < TOL
    if fac == "dense":
        assert so.rel_l2(_num(sp.weight.grad), _num(wc.grad)) < TOL
    else:
        for q, r in zip(sp.weight.parameters(), ref.parameters()):
            assert so.rel_l2(_num(q.grad), _num(r.grad)) < TOL


def test_n_modes_setter_validates_against_the_constructed_block():
    from emu_engine import engine_on_emulation
    from neuraloperator_amd.mpu import SpatialParallelSpectralConv
    with engine_on_emulation():
        sp = SpatialParallelSpectralConv(2, 2, (8, 6))
        assert sp.n_modes == [8, 4] and sp.max_n_modes == [8, 4]
        sp.n_modes = (4, 4)
        assert sp.n_modes == [4, 3] and sp.max_n_modes == [8, 4]
        with pytest.raises(ValueError):
            sp.n_modes = (10, 6)
        with pytest.raises(ValueError):
            sp.n_modes = (4,)
        with pytest.raises(ValueError):
            sp(torch.randn(1, 2, 8, 8, dtype=torch.cfloat))          # complex input on a real layer
