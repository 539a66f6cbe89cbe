"""Time one TFNO (Tucker rank 0.1) SpectralConv layer, fwd+bwd, B=32, C=64, 256^2, modes 64:
native factorized chain vs reconstructed (dense weight rebuilt, dense path)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neuraloperator_amd import SpectralConv
dev = torch.device("cuda:0")
x = torch.randn(32, 64, 256, 256, device=dev, requires_grad=True)
g = torch.randn(32, 64, 256, 256, device=dev)
import sys
for impl in (sys.argv[1:] or ["factorized", "reconstructed"]):
    torch.manual_seed(0)
    conv = SpectralConv(64, 64, (64, 64), factorization="Tucker", rank=0.1, implementation=impl).to(dev)
    def step():
        x.grad = None
        for p in conv.parameters():
            p.grad = None
        conv(x).backward(g)
    for _ in range(int(os.environ.get('TFNO_WARM', '3'))):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n_steps = int(os.environ.get('TFNO_STEPS', '10'))
    for _ in range(n_steps):
        step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / n_steps * 1e3
    print(f"Tucker rank {tuple(conv.weight.core.shape)} implementation={impl}: {ms:.3f} ms/step  {32 / ms * 1e3:.0f} samples/s")
