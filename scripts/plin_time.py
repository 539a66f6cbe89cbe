"""1 x 1 linear skip (sc_pointwise_linear_forward / _backward) at the metric shape, B = 32, 64 channels, 256^2, for
several builds of the engine.  usage: plin_time.py lib.so ..."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neuraloperator_amd import _lib
dev = torch.device("cuda:0")
B, C, S = 32, 64, 256 * 256
torch.manual_seed(0)
x, go = torch.randn(B, C, S, device=dev), torch.randn(B, C, S, device=dev)
w, bias = torch.randn(C, C, device=dev) / 8, torch.randn(C, device=dev)
out, gx = torch.empty_like(x), torch.empty_like(x)
gw, gb = torch.empty_like(w), torch.empty_like(bias)
st = torch.cuda.current_stream().cuda_stream
p = lambda t: t.data_ptr()
libs = [(_p, _lib.ScEngineLib(_p)) for _p in (sys.argv[1:] or [_lib.DEFAULT_LIB])]
for rep in range(3):
    for path, lib in libs:
        ws = torch.empty(max(lib.pointwise_linear_workspace_bytes(B, C, C, S), 256), dtype=torch.uint8, device=dev)
        fw = lambda: lib.pointwise_linear_forward(B, C, C, S, p(x), p(w), p(bias), p(out), st)
        bw = lambda: lib.pointwise_linear_backward(B, C, C, S, p(x), p(w), p(go), p(gx), p(gw), p(gb), p(ws), st)
        res = []
        for fn in (fw, bw):
            for _ in range(10):
                fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                fn()
            e1.record(); torch.cuda.synchronize()
            res.append(e0.elapsed_time(e1) / 20)
        print(f"{os.path.basename(path):28s} forward {res[0]:.3f} ms   backward {res[1]:.3f} ms   checksums {float(out.double().sum()):.6e} {float(gx.double().sum()):.6e}", flush=True)
