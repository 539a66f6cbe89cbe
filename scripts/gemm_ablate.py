"""Time the three layer contractions through the C-ABI, warm and cold (a 600 MB write between
launches), with the MFMA kernel's ablation bits.  Usage: python scripts/gemm_ablate.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neuraloperator_amd import _lib

lib = _lib.get_lib()
dev = torch.device("cuda:0")
B, C, M = 32, 64, 2112
xh = torch.randn(B, C, M, 2, device=dev)
gh = torch.randn(B, C, M, 2, device=dev)
w = torch.randn(C, C, M, 2, device=dev)
out_s = torch.empty(B, C, M, 2, device=dev)
out_w = torch.empty(C, C, M, 2, device=dev)
junk = torch.empty(600 * 1024 * 1024 // 4, device=dev)
st = torch.cuda.current_stream().cuda_stream
kws = {
    "fwd": (xh, w, out_s, dict(P=B, Q=C, R=C, n_modes=M, a_sp=C * M, a_sr=M, a_sm=1, b_sr=C * M, b_sq=M, b_sm=1, c_sp=C * M, c_sq=M, c_sm=1)),
    "gx": (gh, w, out_s, dict(P=B, Q=C, R=C, n_modes=M, a_sp=C * M, a_sr=M, a_sm=1, b_sr=M, b_sq=C * M, b_sm=1, conj_b=1, c_sp=C * M, c_sq=M, c_sm=1)),
    "gw": (xh, gh, out_w, dict(P=C, Q=C, R=B, n_modes=M, a_sp=M, a_sr=C * M, a_sm=1, conj_a=1, b_sr=C * M, b_sq=M, b_sm=1, c_sp=C * M, c_sq=M, c_sm=1)),
}

def t(fn, cold, iters=10):
    fn(); torch.cuda.synchronize()
    tot = 0.0
    for _ in range(iters):
        if cold:
            junk.fill_(1.0)
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        tot += e0.elapsed_time(e1)
    return tot / iters * 1e3

variants = [("full", 0), ("no-mfma", 1 << 24), ("no-store", 2 << 24), ("no-load", 4 << 24),
            ("no-mfma no-store", 3 << 24), ("loads only(no mfma/store)", 3 << 24), ("mfma only", 6 << 24), ("mfma only, no commit/barrier", 14 << 24),
            ("paired (2 WG/CU, 5 modes)", _lib.SC_GEMM_PAIRED), ("VALU kernel", _lib.SC_GEMM_FORCE_VALU)]
for name, (a, b, c, kw) in list(kws.items()):
    for vn, fl in variants:
        fn = lambda: lib.modegemm(a.data_ptr(), b.data_ptr(), c.data_ptr(), st, flags=fl, **kw)
        print(f"{name:4s} {vn:28s} warm {t(fn, False):8.1f} us   cold {t(fn, True):8.1f} us", flush=True)
