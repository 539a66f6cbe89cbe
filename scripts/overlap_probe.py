"""Do the stages of the layer overlap when launched on two streams?  (read-bound forward transform,
write-bound inverse transform, short contractions.)  Times pairs sequentially on one stream and
concurrently on two.  Usage: python scripts/overlap_probe.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neuraloperator_amd import _lib
lib = _lib.get_lib()
dev = torch.device("cuda:0")
B, C, H = 32, 64, 256
M = 64 * 33
plan = lib.plan_create([H, 256], [64, 33])


def bufs(b):
    return dict(x=torch.randn(b, C, H, 256, device=dev), y=torch.empty(b, C, H, 256, device=dev),
                xh=torch.randn(b, C, M, 2, device=dev), yh=torch.randn(b, C, M, 2, device=dev),
                oh=torch.empty(b, C, M, 2, device=dev))


w = torch.randn(C, C, M, 2, device=dev)
gw = torch.empty(C, C, M, 2, device=dev)
bias = torch.randn(C, device=dev)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def fwd(d, b, st): lib.transform_forward(plan, 0, d["x"].data_ptr(), d["xh"].data_ptr(), b * C, 0, st.cuda_stream)
def inv(d, b, st): lib.transform_inverse(plan, 0, d["yh"].data_ptr(), bias.data_ptr(), C, d["y"].data_ptr(), b * C, 0, st.cuda_stream)
def gemm(d, b, st):
    lib.modegemm(d["xh"].data_ptr(), w.data_ptr(), d["oh"].data_ptr(), st.cuda_stream, P=b, Q=C, R=C, n_modes=M,
                 a_sp=C * M, a_sr=M, a_sm=1, b_sr=C * M, b_sq=M, b_sm=1, c_sp=C * M, c_sq=M, c_sm=1)
def gemm_gw(d, b, st):
    lib.modegemm(d["xh"].data_ptr(), d["yh"].data_ptr(), gw.data_ptr(), st.cuda_stream, P=C, Q=C, R=b, n_modes=M,
                 a_sp=M, a_sr=C * M, a_sm=1, b_sr=C * M, b_sq=M, b_sm=1, c_sp=C * M, c_sq=M, c_sm=1, conj_a=1,
                 flags=_lib.SC_GEMM_STREAM_C)


def timed(fa, fb, concurrent, n=20):
    def once():
        if concurrent:
            fa(s1); fb(s2)
        else:
            fa(s1); fb(s1)
    for _ in range(3):
        once()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    s1.wait_stream(torch.cuda.current_stream()); s2.wait_stream(torch.cuda.current_stream())
    e0.record(torch.cuda.current_stream())
    s1.wait_event(e0); s2.wait_event(e0)
    for _ in range(n):
        once()
    torch.cuda.current_stream().wait_stream(s1); torch.cuda.current_stream().wait_stream(s2)
    e1.record(torch.cuda.current_stream())
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


d1, d2 = bufs(B), bufs(B)
h1, h2 = bufs(B // 2), bufs(B // 2)
pairs = [
    ("fwd(B) + inv(B)", lambda s: fwd(d1, B, s), lambda s: inv(d2, B, s)),
    ("fwd(B) + fwd(B)", lambda s: fwd(d1, B, s), lambda s: fwd(d2, B, s)),
    ("inv(B) + inv(B)", lambda s: inv(d1, B, s), lambda s: inv(d2, B, s)),
    ("gemm gW + inv(B)", lambda s: gemm_gw(d1, B, s), lambda s: inv(d2, B, s)),
    ("gemm gW + gemm gX", lambda s: gemm_gw(d1, B, s), lambda s: gemm(d2, B, s)),
    ("gemm + fwd(B)", lambda s: gemm(d1, B, s), lambda s: fwd(d2, B, s)),
    ("fwd(B/2) + inv(B/2)", lambda s: fwd(h1, B // 2, s), lambda s: inv(h2, B // 2, s)),
    ("fwd(B/2) + fwd(B/2)", lambda s: fwd(h1, B // 2, s), lambda s: fwd(h2, B // 2, s)),
]
for name, fa, fb in pairs:
    ts, tc = timed(fa, fb, False), timed(fa, fb, True)
    print(f"{name:24s} sequential {ts:7.1f} us   two streams {tc:7.1f} us   ({tc / ts:.2f}x)", flush=True)
