"""Time the fused FFT transforms (event-timed, in a forward/inverse sequence) for a given engine .so.
Usage: python scripts/fft_time.py [path/to/libsc_engine.so] [flags]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neuraloperator_amd import _lib
path = sys.argv[1] if len(sys.argv) > 1 else _lib.DEFAULT_LIB
flags = int(sys.argv[2], 0) if len(sys.argv) > 2 else 0
lib = _lib.ScEngineLib(path)
dev = torch.device("cuda:0")
B, C, H = 32, 64, 256
plan = lib.plan_create([H, 256], [64, 33], flags=flags)
x = torch.randn(B, C, H, 256, device=dev); y = torch.empty_like(x)
xh = torch.randn(B, C, 64 * 33, 2, device=dev)
bias = torch.randn(C, device=dev)
st = torch.cuda.current_stream().cuda_stream
f = lambda: lib.transform_forward(plan, 0, x.data_ptr(), xh.data_ptr(), B * C, 0, st)
i = lambda: lib.transform_inverse(plan, 0, xh.data_ptr(), bias.data_ptr(), C, y.data_ptr(), B * C, 0, st)
for _ in range(3):
    f(); i()
torch.cuda.synchronize()
tf = ti = 0.0
n = 10
for _ in range(n):
    e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    e[0].record(); f(); e[1].record(); i(); e[2].record(); torch.cuda.synchronize()
    tf += e[0].elapsed_time(e[1]); ti += e[1].elapsed_time(e[2])
print(f"{os.path.basename(path)} flags={flags} kernel {lib.plan_kernel_name(plan, 0)}: fwd {tf / n * 1e3:.1f} us  inv {ti / n * 1e3:.1f} us")
