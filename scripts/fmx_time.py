"""Time the factor-matrix product and the mode-summed contraction of the TFNO chain alone (configs[2] shapes)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neuraloperator_amd import engine
dev = torch.device("cuda:0")
B, C, Rk, M = 32, 64, 36, 2112
c = lambda *s: torch.randn(*s, dtype=torch.complex64, device=dev)
xhat, u, z, gz = c(B, C, M), c(C, Rk), c(B, Rk, M), c(B, Rk, M)
def t(name, fn):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): fn()
    e1.record(); torch.cuda.synchronize()
    print(f"{os.environ.get('TAG', '')} {name}: {e0.elapsed_time(e1) / 50 * 1e3:.1f} us")
t("z = xhat U (R=64, Q=36)", lambda: engine._raw_mode_gemm(xhat, u, M, False, False))
t("yhat = t U^T (R=36, Q=64)", lambda: engine._raw_mode_gemm(z, u.transpose(0, 1), M, False, False))
t("gU = sum xhat^H gz (P=64, Q=36)", lambda: engine._raw_mode_gemm(xhat.transpose(0, 1), gz, M, True, False, reduce_modes=True))
