"""Launch the fused FFT transforms a few times (for rocprofv3 --pmc / --kernel-trace).
Usage: python scripts/fft_one.py [bf16]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neuraloperator_amd import _lib
from neuraloperator_amd.engine import get_plan
lib = _lib.get_lib()
dev = torch.device("cuda:0")
B, C, H = 32, 64, 256
bf16 = len(sys.argv) > 1 and sys.argv[1] == "bf16"          # SC_PLAN_IO_BF16: real tensors as bfloat16
plan = get_plan(dev, [H, 256], [64, 33], "forward", _lib.SC_PLAN_IO_BF16 if bf16 else 0)
x = torch.randn(B, C, H, 256, device=dev).to(torch.bfloat16 if bf16 else torch.float32); y = torch.empty_like(x)
xh = torch.randn(B, C, 64 * 33, 2, device=dev)
bias = torch.randn(C, device=dev)
st = torch.cuda.current_stream().cuda_stream
for _ in range(5):
    lib.transform_forward(plan, 0, x.data_ptr(), xh.data_ptr(), B * C, 0, st)
    lib.transform_inverse(plan, 0, xh.data_ptr(), bias.data_ptr(), C, y.data_ptr(), B * C, 0, st)
torch.cuda.synchronize()
