"""rocprofv3 --pmc / --kernel-trace target: the whole layer step (forward + backward through the C-ABI) of the
metric shape, 5 times.  Every kernel of the step appears with its in-step cache state."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from engine_runner import layer_fwd_bwd  # noqa: E402
from neuraloperator_amd import _lib  # noqa: E402

lib = _lib.get_lib()
dev = torch.device("cuda:0")
# LAYER_SHAPE = "B,C,spatial...,modes..." (default: the metric shape 32,64,256,256,64,64)
shape = [int(v) for v in os.environ.get("LAYER_SHAPE", "32,64,256,256,64,64").split(",")]
B, C = shape[:2]
nd = (len(shape) - 2) // 2
spatial, modes = shape[2:2 + nd], shape[2 + nd:]
kept = modes[:-1] + [modes[-1] // 2 + 1]
torch.manual_seed(0)
x = torch.randn(B, C, *spatial, device=dev)
g = torch.randn(B, C, *spatial, device=dev)
w = torch.randn(C, C, *kept, dtype=torch.cfloat, device=dev)
bias = torch.randn(C, *([1] * nd), device=dev)
for _ in range(int(os.environ.get("LAYER_REPS", 5))):
    layer_fwd_bwd(lib, x, w, bias, g, kept, kept)
torch.cuda.synchronize()
