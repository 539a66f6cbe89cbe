"""A-B of the contraction kernels between engine builds, same box, interleaved:
python scripts/gemm_ab.py libA.so libB.so ...   (us per launch, warm = back to back, cold = after a 600 MB fill)"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neuraloperator_amd import _lib
paths = sys.argv[1:] or [_lib.DEFAULT_LIB]
libs = [(_os.path.basename(p), _lib.ScEngineLib(p)) for _os in [os] for p in paths]
dev = torch.device("cuda:0")
M = 2112
st = torch.cuda.current_stream().cuda_stream
junk = torch.empty(600 * 1024 * 1024 // 4, device=dev)


def shapes():
    # name, P, Q, R, A dims, B dims (None = mode independent), conj_a, conj_b
    return [("fwd  P32 Q64 R64", 32, 64, 64, True), ("gW   P64 Q64 R32", 64, 64, 32, True),
            ("z    P32 Q36 R64 (B mode-indep.)", 32, 36, 64, False), ("t    P32 Q36 R36", 32, 36, 36, True),
            ("gT   P36 Q36 R32", 36, 36, 32, True)]


def timed(fn, cold, n=10):
    tot = 0.0
    for _ in range(n):
        if cold:
            junk.fill_(1.0)
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        tot += e0.elapsed_time(e1)
    return tot / n * 1e3


for name, P, Q, R, bmode in shapes():
    a = torch.randn(P, R, M, 2, device=dev)
    b = torch.randn(R, Q, M, 2, device=dev) if bmode else torch.randn(R, Q, 2, device=dev)
    c = torch.empty(P, Q, M, 2, device=dev)
    kw = dict(P=P, Q=Q, R=R, n_modes=M, a_sp=R * M, a_sr=M, a_sm=1, c_sp=Q * M, c_sq=M, c_sm=1)
    kw.update(dict(b_sr=Q * M, b_sq=M, b_sm=1) if bmode else dict(b_sr=Q, b_sq=1, b_sm=0))
    row = [f"{name:36s}"]
    for lname, lib in libs:
        fn = lambda: lib.modegemm(a.data_ptr(), b.data_ptr(), c.data_ptr(), st, **kw)
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        row.append(f"{lname}: mfma={int(lib.modegemm_uses_matrix_cores(**kw))} warm {timed(fn, False):6.1f} cold {timed(fn, True):6.1f}")
    print(" | ".join(row), flush=True)
