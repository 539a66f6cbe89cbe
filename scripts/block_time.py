"""One FNO block at the metric shape, forward + backward: the reference's op sequence
(spectral conv on the engine + PyTorch for everything around it) against the two fused engine passes
(neuraloperator_amd.blocks.fused_block_forward).  Usage: python scripts/block_time.py [B C H W modes]"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from block_standin import Blocks
from neuraloperator_amd import blocks as nb

a = [int(v) for v in sys.argv[1:]] or [32, 64, 256, 256, 64]
B, C, H, W, M = a
dev = torch.device("cuda:0")
torch.manual_seed(0)
blk = Blocks(C, (M, M)).to(dev)
x = torch.randn(B, C, H, W, device=dev, requires_grad=True)
g = torch.randn(B, C, H, W, device=dev)


def step(fn):
    blk.zero_grad(set_to_none=True)
    x.grad = None
    y = fn(x)
    y.backward(g)
    return y


def timeit(fn, fwd_only=False, n=10):
    for _ in range(3):
        (fn(x) if fwd_only else step(fn))
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        (fn(x) if fwd_only else step(fn))
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


unf = lambda t: blk(t, 0)
fus = lambda t: nb.fused_block_forward(blk, t, 0)
if os.environ.get("BLOCK_ONLY_FUSED"):                     # kernel-trace target: only the fused path's launches
    for _ in range(5):
        step(fus)
    torch.cuda.synchronize()
    sys.exit(0)
assert nb._block_in_scope(blk, 0, None)
y0, y1 = unf(x), fus(x)
print(f"agreement fused vs op sequence: rel-L2 {((y1 - y0).norm() / y0.norm()).item():.2e}")
R = B * C * H * W * 4 / 1e6
print(f"block B={B} C={C} {H}x{W} modes {M} (one activation tensor R = {R:.0f} MB)")
with torch.no_grad():
    tf1 = timeit(fus, True)
t1 = timeit(fus)
if C < 128 or os.environ.get("BLOCK_REF"):               # (F.conv1d's fp32 backward at 128 channels: MIOpen's naive kernels, 338 ms per call)
    with torch.no_grad():
        tf0 = timeit(unf, True)
    t0 = timeit(unf)
    print(f"  reference op sequence (engine conv + PyTorch glue): forward {tf0:.3f} ms, forward+backward {t0:.3f} ms")
print(f"  fused (Fourier-layer epilogue + pointwise MLP pass):  forward {tf1:.3f} ms, forward+backward {t1:.3f} ms")
