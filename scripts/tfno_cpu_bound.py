"""Is the TFNO step host-bound?  Host time to ISSUE a step (no synchronisation) against the device time per step."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neuraloperator_amd import SpectralConv
dev = torch.device("cuda:0")
x = torch.randn(32, 64, 256, 256, device=dev, requires_grad=True)
g = torch.randn(32, 64, 256, 256, device=dev)
torch.manual_seed(0)
conv = SpectralConv(64, 64, (64, 64), factorization="tucker", rank=0.1, implementation="factorized").to(dev)
def step():
    x.grad = None
    for p in conv.parameters():
        p.grad = None
    conv(x).backward(g)
for _ in range(40):
    step()
torch.cuda.synchronize()
for rep in range(3):
    t0 = time.perf_counter()
    for _ in range(20):
        step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"issue {1e3 * (t1 - t0) / 20:.3f} ms/step, complete {1e3 * (t2 - t0) / 20:.3f} ms/step")
