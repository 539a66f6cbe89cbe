"""rocprofv3 --pmc / --kernel-trace target: the whole layer step (forward + backward) of one workload, LAYER_REPS
times.  Every kernel of the step appears with its in-step cache state.

    LAYER_SHAPE = "B,C,spatial...,modes..."   (default: the metric shape 32,64,256,256,64,64)
    LAYER_IO    = f32 | bf16                  bf16: SC_PLAN_IO_BF16 real tensors (BASELINE configs[1])
    LAYER_KIND  = dense | tucker              dense: sc_layer_forward / _backward through the C-ABI (tests/engine_runner);
                                              tucker: the drop-in module with Tucker rank 0.1, factorized (configs[2])
"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from engine_runner import layer_fwd_bwd  # noqa: E402
from neuraloperator_amd import _lib  # noqa: E402

lib = _lib.get_lib()
dev = torch.device("cuda:0")
shape = [int(v) for v in os.environ.get("LAYER_SHAPE", "32,64,256,256,64,64").split(",")]
B, C = shape[:2]
nd = (len(shape) - 2) // 2
spatial, modes = shape[2:2 + nd], shape[2 + nd:]
kept = modes[:-1] + [modes[-1] // 2 + 1]
io = os.environ.get("LAYER_IO", "f32")
kind = os.environ.get("LAYER_KIND", "dense")
reps = int(os.environ.get("LAYER_REPS", 5))
dt = torch.bfloat16 if io == "bf16" else torch.float32
torch.manual_seed(0)
x = torch.randn(B, C, *spatial, device=dev).to(dt)
g = torch.randn(B, C, *spatial, device=dev).to(dt)
if kind == "tucker":
    from neuraloperator_amd import SpectralConv
    conv = SpectralConv(C, C, tuple(modes), factorization="tucker", rank=0.1, implementation="factorized").to(dev)
    x.requires_grad_(True)
    for _ in range(reps):
        x.grad = None
        for q in conv.parameters():
            q.grad = None
        conv(x).backward(g)
else:
    w = torch.randn(C, C, *kept, dtype=torch.cfloat, device=dev)
    bias = torch.randn(C, *([1] * nd), device=dev)
    flags = _lib.SC_PLAN_IO_BF16 if io == "bf16" else 0
    for _ in range(reps):
        layer_fwd_bwd(lib, x, w, bias, g, kept, kept, flags=flags)
torch.cuda.synchronize()
