"""CPU tier: fno_block_precision "half" / "mixed" on the engine (host emulation).

* sc_round_f16 against torch's float32 -> float16 cast, bit for bit (normals, subnormals, overflow, specials);
* sc_modegemm with SC_GEMM_F16 against the oracle's restatement of ``einsum_complexhalf_two_input``
  (einsum_utils.py:10-36; pinned against the verbatim function in tests/test_oracle_vs_reference.py);
* the module's half / mixed forward + backward (neuraloperator_amd.SpectralConv._forward_half) against the golden
  vectors of oracle.gen_golden.gen_half."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from emu_engine import engine_on_emulation
from engine_runner import emu_lib, rel_l2
from neuraloperator_amd import _lib
from oracle import spectral_oracle as so


@pytest.fixture(scope="module")
def lib():
    return emu_lib()


def test_round_f16_is_torchs_cast(lib):
    rng = np.random.default_rng(0)
    v = np.concatenate([
        rng.standard_normal(4096).astype(np.float32) * np.float32(10.0) ** rng.integers(-9, 6, 4096).astype(np.float32),
        np.array([0.0, -0.0, 65504.0, 65519.9, 65520.0, -65520.0, 1e9, 5.96e-8, 2.98e-8, 2.99e-8, 8.94e-8, 6.1e-5,
                  6.09e-5, np.inf, -np.inf], dtype=np.float32),
        np.float32(1.0) + np.arange(0, 64, dtype=np.float32) * np.float32(2.0 ** -13),       # ties of the 10-bit mantissa
    ]).astype(np.float32)
    t = torch.from_numpy(v.copy())
    out = torch.empty_like(t)
    lib.round_f16(t.data_ptr(), out.data_ptr(), t.numel(), 0)
    ref = t.half().float()
    assert torch.equal(out.view(torch.int32), ref.view(torch.int32))
    lib.round_f16(t.data_ptr(), t.data_ptr(), t.numel(), 0)                                   # in place
    assert torch.equal(t.view(torch.int32), ref.view(torch.int32))


@pytest.mark.parametrize("conj_a,conj_b", [(0, 0), (1, 0), (0, 1)])
@pytest.mark.parametrize("shape", [(3, 5, 6, 70), (8, 32, 32, 64), (2, 17, 3, 9)], ids=str)
def test_f16_contraction_is_einsum_complexhalf(lib, shape, conj_a, conj_b):
    P, R, Q, M = shape
    g = torch.Generator().manual_seed(11)
    a = torch.randn(P, R, M, dtype=torch.complex64, generator=g)
    b = torch.randn(R, Q, M, dtype=torch.complex64, generator=g) * 0.3
    c = torch.full((P, Q, M), float("nan"), dtype=torch.complex64)
    lib.modegemm(torch.view_as_real(a).data_ptr(), torch.view_as_real(b).data_ptr(), torch.view_as_real(c).data_ptr(), 0,
                 P=P, Q=Q, R=R, n_modes=M, a_sp=R * M, a_sr=M, a_sm=1, b_sr=Q * M, b_sq=M, b_sm=1,
                 c_sp=Q * M, c_sq=M, c_sm=1, conj_a=conj_a, conj_b=conj_b, flags=_lib.SC_GEMM_F16)
    ref = so.contract_dense_chalf((a.conj() if conj_a else a).resolve_conj(), (b.conj() if conj_b else b).resolve_conj())
    got, ref = torch.view_as_real(c), torch.view_as_real(ref)
    assert torch.equal(got.half().float(), got), "float16-representable values"
    same = (got == ref).float().mean().item()
    ulp = torch.maximum(ref.abs(), torch.tensor(6.1e-5)) * 2.0 ** -10
    assert same > 0.999 and bool(((got - ref).abs() <= ulp).all()), f"{same:.5f} bit-identical"


@pytest.mark.parametrize("name", ["half_2d", "mixed_2d", "mixed_3d"])
def test_module_half_precision_matches_golden(name):
    from neuraloperator_amd import SpectralConv
    g = load_golden(name)
    x = torch.from_numpy(g["x"]).requires_grad_(True)
    ci, co = g["w"].shape[:2]
    conv = SpectralConv(ci, co, tuple(int(v) for v in g["ctor_n_modes"]), fno_block_precision=str(g["precision"]))
    with torch.no_grad():
        conv.weight.tensor.copy_(torch.from_numpy(g["w"]))
        conv.bias.copy_(torch.from_numpy(g["bias"]))
    with engine_on_emulation():
        y = conv._forward_half(x, list(x.shape[2:]))
        y.backward(torch.from_numpy(g["g"]))
    ref = torch.from_numpy(g["y"])
    # y = float16(inverse transform) + bias: the fp32 transforms differ from torch's in the last bits, which moves a
    # value across a float16 rounding boundary now and then -- never by more than one float16 step of the pre-bias value
    yb = (y.detach() - conv.bias.detach())
    rb = ref - conv.bias.detach()
    step = torch.maximum(rb.abs(), torch.tensor(6.1e-5)) * 2.0 ** -10
    assert bool(((yb - rb).abs() <= 1.01 * step).all())
    assert ((yb - rb).abs() <= 1e-7).float().mean().item() > 0.98
    # gradients: the reference differentiates its float16 einsum in float16 (one rounding per gradient element), the
    # engine's gradient contractions round the four real products and their combination: same values to ~2 float16 ulp
    assert rel_l2(x.grad.numpy(), g["gx"]) < 2e-3
    assert rel_l2(conv.weight.tensor.grad.numpy(), g["gw"]) < 2e-3
    assert rel_l2(conv.bias.grad.numpy(), g["gbias"]) < 1e-5
