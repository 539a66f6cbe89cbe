"""Small layers are host-bound (python + autograd dispatch): the same step captured once into a HIP graph
and replayed.  Usage: python scripts/graph_time.py"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neuraloperator_amd import SpectralConv
dev = torch.device("cuda:0")
for (B, C, spatial, modes) in [(4, 32, (16, 16), (12, 12)), (64, 64, (64, 64), (32, 32)), (32, 64, (256, 256), (64, 64))]:
    conv = SpectralConv(C, C, modes).to(dev)
    x = torch.randn(B, C, *spatial, device=dev, requires_grad=True)
    g = torch.randn(B, C, *spatial, device=dev)

    def step():
        y = conv(x)
        return torch.autograd.grad(y, (x, conv.weight.tensor, conv.bias), g)

    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3):
            step()
    torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()

    def timed(fn, n=50):
        fn(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3

    eager = timed(step)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        out = step()
    replay = timed(graph.replay)
    print(f"B={B} C={C} {spatial} modes {modes}: eager {eager:.3f} ms/step   graph replay {replay:.3f} ms/step", flush=True)
