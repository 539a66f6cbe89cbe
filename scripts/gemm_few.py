"""4-mode vs 9-mode workgroups of the matrix-core contraction on small mode counts (us per launch, warm)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neuraloperator_amd import _lib
lib = _lib.get_lib()
dev = torch.device("cuda:0")
st = torch.cuda.current_stream().cuda_stream


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for (P, Q, R, M) in [(64, 64, 64, 544), (32, 64, 64, 544), (32, 64, 64, 144), (64, 64, 64, 1056), (32, 64, 64, 2112)]:
    a = torch.randn(P, R, M, 2, device=dev); b = torch.randn(R, Q, M, 2, device=dev); c = torch.empty(P, Q, M, 2, device=dev)
    kw = dict(P=P, Q=Q, R=R, n_modes=M, a_sp=R * M, a_sr=M, a_sm=1, b_sr=Q * M, b_sq=M, b_sm=1, c_sp=Q * M, c_sq=M, c_sm=1)
    row = [f"P{P} Q{Q} R{R} M{M}:"]
    for name, fl in (("auto", 0), ("wide9", _lib.SC_GEMM_WIDE), ("valu", _lib.SC_GEMM_FORCE_VALU)):
        row.append(f"{name} {timed(lambda: lib.modegemm(a.data_ptr(), b.data_ptr(), c.data_ptr(), st, flags=fl, **kw)):6.1f}")
    print("  ".join(row), flush=True)
