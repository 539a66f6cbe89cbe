"""Pointwise MLP backward (metric shape, 64 -> 32 -> 64) with and without the folded GELU backward of the Fourier layer
(x_pre), for several builds of the engine.  usage: pmlp_xpre_time.py lib.so ..."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neuraloperator_amd import _lib
dev = torch.device("cuda:0")
B, C, Hd, S = 32, 64, 32, 256 * 256
torch.manual_seed(0)
x, sk, go, xp = (torch.randn(B, C, S, device=dev) for _ in range(4))
w1, b1, w2, b2, gt = torch.randn(Hd, C, device=dev) / 8, torch.randn(Hd, device=dev), torch.randn(C, Hd, device=dev) / 6, torch.randn(C, device=dev), torch.randn(C, device=dev)
gx, gsk = torch.empty_like(x), torch.empty_like(x)
gw1, gw2, gb1, gb2, gg = torch.empty_like(w1), torch.empty_like(w2), torch.empty_like(b1), torch.empty_like(b2), torch.empty_like(gt)
st = torch.cuda.current_stream().cuda_stream
p = lambda t: t.data_ptr()
libs = [(_p, _lib.ScEngineLib(_p)) for _p in (sys.argv[1:] or [_lib.DEFAULT_LIB])]
for rep in range(3):
    for path, lib in libs:
        ws = torch.empty(lib.pointwise_mlp_workspace_bytes(B, C, Hd, C, S, 1), dtype=torch.uint8, device=dev)
        res = []
        for xpre in (0, p(xp)):
            bw = lambda: lib.pointwise_mlp_backward(B, C, Hd, C, S, 1, p(x), p(w1), p(b1), p(w2), p(b2), p(sk), p(gt), p(go), p(gx),
                                                    p(gw1), p(gb1), p(gw2), p(gb2), p(gsk), p(gg), p(ws), st, x_pre=xpre)
            for _ in range(10):
                bw()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                bw()
            e1.record(); torch.cuda.synchronize()
            res.append(e0.elapsed_time(e1) / 20)
        print(f"{os.path.basename(path):28s} backward {res[0]:.3f} ms   with x_pre {res[1]:.3f} ms   checksum {float(gx.double().sum()):.6e} {float(gw1.double().sum()):.6e}", flush=True)
