"""TFNO rank 0.1 step (BASELINE configs[2]) eager against torch.cuda.make_graphed_callables (hipGraph replay of the
same launches): is the eager step host-bound on this box?"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neuraloperator_amd import SpectralConv
dev = torch.device("cuda:0")
torch.manual_seed(0)
x = torch.randn(32, 64, 256, 256, device=dev, requires_grad=True)
g = torch.randn(32, 64, 256, 256, device=dev)
kw = dict(factorization="tucker", rank=0.1, implementation="factorized") if "dense" not in sys.argv else {}
conv = SpectralConv(64, 64, (64, 64), **kw).to(dev)


def timeit(step, n=40):
    for _ in range(60):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    return 1e3 * (t1 - t0) / n, 1e3 * (t2 - t0) / n


def eager():
    x.grad = None
    for p in conv.parameters():
        p.grad = None
    conv(x).backward(g)


print("eager   issue %.3f ms  complete %.3f ms" % timeit(eager))
y_ref = conv(x).detach().clone()
try:
    gconv = torch.cuda.make_graphed_callables(conv, (x.detach().clone().requires_grad_(True),))
    xs = x.detach().clone().requires_grad_(True)

    def graphed():
        xs.grad = None
        for p in conv.parameters():
            p.grad = None
        gconv(xs).backward(g)

    print("graphed issue %.3f ms  complete %.3f ms" % timeit(graphed))
    y2 = gconv(xs).detach()
    print("graphed vs eager output rel-L2:", ((y2 - y_ref).norm() / y_ref.norm()).item())
except Exception as e:  # noqa: BLE001
    print("graph capture failed:", type(e).__name__, str(e)[:300])
