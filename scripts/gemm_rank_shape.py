"""The three contractions of ONE rank of an 8-rank mode-parallel FNO3d 128^3 layer (BASELINE configs[3]): 8 samples x
32 x 32 channels x 2176 modes (4 of 32 first-dim rows x 32 x 17) -- us per launch, warm, by route (environment SC_SB_MAX
is read once per process: run the script once per value)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neuraloperator_amd import _lib
lib = _lib.get_lib()
dev = torch.device("cuda:0")
st = torch.cuda.current_stream().cuda_stream
B, Ci, Co, M = 8, 32, 32, 2176


def timed(fn, n=50):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


xh = torch.randn(B, Ci, M, 2, device=dev); gh = torch.randn(B, Co, M, 2, device=dev); w = torch.randn(Ci, Co, M, 2, device=dev)
y = torch.empty(B, Co, M, 2, device=dev); gw = torch.empty(Ci, Co, M, 2, device=dev); gx = torch.empty(B, Ci, M, 2, device=dev)
kw_f = dict(P=B, Q=Co, R=Ci, n_modes=M, a_sp=Ci * M, a_sr=M, a_sm=1, b_sr=Co * M, b_sq=M, b_sm=1, c_sp=Co * M, c_sq=M, c_sm=1)
kw_w = dict(P=Ci, Q=Co, R=B, n_modes=M, a_sp=M, a_sr=Ci * M, a_sm=1, conj_a=1, b_sr=Co * M, b_sq=M, b_sm=1, c_sp=Co * M, c_sq=M, c_sm=1)
kw_x = dict(P=B, Q=Ci, R=Co, n_modes=M, a_sp=Co * M, a_sr=M, a_sm=1, b_sr=M, b_sq=Co * M, b_sm=1, conj_b=1, c_sp=Ci * M, c_sq=M, c_sm=1)
print(f"SC_SB_MAX={os.environ.get('SC_SB_MAX', '(default 4)')}")
for name, fl in (("auto", 0), ("no_sb", _lib.SC_GEMM_NO_SB), ("valu", _lib.SC_GEMM_FORCE_VALU | _lib.SC_GEMM_NO_SB), ("sb_alt", _lib.SC_GEMM_SB_ALT_ORDER)):
    tf = timed(lambda: lib.modegemm(xh.data_ptr(), w.data_ptr(), y.data_ptr(), st, flags=fl, **kw_f))
    tw = timed(lambda: lib.modegemm(xh.data_ptr(), gh.data_ptr(), gw.data_ptr(), st, flags=fl, **kw_w))
    tx = timed(lambda: lib.modegemm(gh.data_ptr(), w.data_ptr(), gx.data_ptr(), st, flags=fl, **kw_x))
    tp = timed(lambda: lib.modegemm_pair(dict(kw_w, flags=fl), xh.data_ptr(), gh.data_ptr(), gw.data_ptr(), dict(kw_x, flags=fl), gh.data_ptr(), w.data_ptr(), gx.data_ptr(), st))
    print(f"{name:7s} paths {lib.modegemm_path(**dict(kw_f, flags=fl))} {lib.modegemm_path(**dict(kw_w, flags=fl))} {lib.modegemm_path(**dict(kw_x, flags=fl))}   fwd {tf:6.1f}  gW {tw:6.1f}  gx {tx:6.1f}  pair {tp:6.1f} us", flush=True)
