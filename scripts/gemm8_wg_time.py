"""How long ONE workgroup of the streamed contraction kernel (k_modegemm_dma, narrow shape) takes: launches of 32 x 32
tiles, 8 modes per workgroup, with G = modes / 8 workgroups and reduction length R -- time against R (stages of 4 r) and
against G (1 / 3 / 9 workgroups per compute unit).  Events on the launch stream, 200 launches back to back."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neuraloperator_amd import _lib
lib = _lib.get_lib()
dev = torch.device("cuda:0")
st = torch.cuda.current_stream().cuda_stream


def run(P, Q, R, M, n=200):
    a = torch.randn(P, R, M, 2, device=dev)
    b = torch.randn(R, Q, M, 2, device=dev)
    c = torch.empty(P, Q, M, 2, device=dev)
    kw = dict(P=P, Q=Q, R=R, n_modes=M, a_sp=R * M, a_sr=M, a_sm=1, b_sr=Q * M, b_sq=M, b_sm=1, c_sp=Q * M, c_sq=M, c_sm=1,
              conj_a=0, conj_b=0, flags=0)
    path = lib.modegemm_path(**kw)
    for _ in range(20):
        lib.modegemm(a.data_ptr(), b.data_ptr(), c.data_ptr(), st, **kw)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(n):
        lib.modegemm(a.data_ptr(), b.data_ptr(), c.data_ptr(), st, **kw)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / n
    mb = (a.numel() + b.numel() + c.numel()) * 4 / 1e6
    return us, mb, path


for wgs in (256, 768, 2304):
    M = wgs * 8
    for (P, Q) in ((32, 32), (64, 64)):
        tiles = (P // 32) * (Q // 32)
        if wgs * tiles > 2304 * 2:
            continue
        for R in (4, 8, 16, 32, 64, 128):
            us, mb, path = run(P, Q, R, M // tiles if tiles > 1 else M)
            print(f"workgroups {wgs:5d}  tile grid {P}x{Q}  modes {M // tiles if tiles > 1 else M:6d}  R {R:4d}  path {path}  {us:7.1f} us  {mb:8.1f} MB  {mb / us / 1e3 * 1e3:6.2f} GB/ms", flush=True)
