"""SURVEY.md 8 row f1, first step: time of the Fourier layer of an FNO block, out = gelu(conv(x) + skip), forward and
forward + backward at the metric shape -- unfused (conv, then ATen add and gelu) vs the fused epilogue
(SpectralConv.forward_fused: addition + activation in the inverse transform's store path)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neuraloperator_amd import SpectralConv  # noqa: E402

dev = torch.device("cuda:0")
B, C, N = 32, 64, 256
conv = SpectralConv(C, C, (64, 64)).to(dev)
x = torch.randn(B, C, N, N, device=dev, requires_grad=True)
skip = torch.randn(B, C, N, N, device=dev, requires_grad=True)
g = torch.randn(B, C, N, N, device=dev)
R = 4 * B * C * N * N


def timed(fn, n=15):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def fwd(fused):
    with torch.no_grad():
        return conv.forward_fused(x, skip, "gelu") if fused else torch.nn.functional.gelu(conv(x) + skip)


def fwd_bwd(fused):
    x.grad = skip.grad = None
    conv.zero_grad(set_to_none=True)
    out = conv.forward_fused(x, skip, "gelu") if fused else torch.nn.functional.gelu(conv(x) + skip)
    out.backward(g)


for name, fn in (("forward", fwd), ("forward + backward", fwd_bwd)):
    a, b = timed(lambda: fn(False)), timed(lambda: fn(True))
    print(f"{name:20s} unfused {a:7.3f} ms   fused epilogue {b:7.3f} ms   ({a - b:+.3f} ms, R = {R / 1e6:.0f} MB)", flush=True)
