"""How long after start-up does the step time settle?  The module step, 600 steps back to back from a cold start,
event-timed in blocks of 10; and again after 2 s of host-side idling."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neuraloperator_amd import SpectralConv

dev = torch.device("cuda:0")
conv = SpectralConv(64, 64, (64, 64)).to(dev)
x = torch.randn(32, 64, 256, 256, device=dev, requires_grad=True)
g = torch.randn(32, 64, 256, 256, device=dev)


def step():
    x.grad = None
    for p in conv.parameters():
        p.grad = None
    conv(x).backward(g)


def series(tag, nblk=60, per=10):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(nblk + 1)]
    ev[0].record()
    for i in range(nblk):
        for _ in range(per):
            step()
        ev[i + 1].record()
    torch.cuda.synchronize()
    ms = [ev[i].elapsed_time(ev[i + 1]) / per for i in range(nblk)]
    print(tag, " ".join(f"{v:.3f}" for v in ms))


series("cold start, ms/step per block of 10 steps:")
time.sleep(2.0)
series("after 2 s idle:")
time.sleep(0.2)
series("after 0.2 s idle:", nblk=20)
