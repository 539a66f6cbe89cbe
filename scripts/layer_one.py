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
B, C, H = 32, 64, 256
torch.manual_seed(0)
x = torch.randn(B, C, H, 256, device=dev)
g = torch.randn(B, C, H, 256, device=dev)
w = torch.randn(C, C, 64, 33, dtype=torch.cfloat, device=dev)
bias = torch.randn(C, 1, 1, device=dev)
for _ in range(5):
    layer_fwd_bwd(lib, x, w, bias, g, [64, 33], [64, 33])
torch.cuda.synchronize()
