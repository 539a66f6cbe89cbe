"""One SpectralConv fwd + bwd step eager against a hipGraph replay of the same launches (torch.cuda.CUDAGraph around
conv(x).backward(g) with static x / g: no input copies), for workloads whose eager step is bound by the host's issue
rate.  usage: graph_step_time.py <workload> [...]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neuraloperator_amd import SpectralConv
from bench import WORKLOADS
dev = torch.device("cuda:0")


def timeit(step, n=50):
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


for wl in sys.argv[1:]:
    B, C, spatial, n_modes = WORKLOADS[wl]
    torch.manual_seed(0)
    x = torch.randn(B, C, *spatial, device=dev, requires_grad=True)
    g = torch.randn(B, C, *spatial, device=dev)
    conv = SpectralConv(C, C, n_modes).to(dev)
    params = list(conv.parameters())

    def eager():
        x.grad = None
        for p in params:
            p.grad = None
        conv(x).backward(g)

    ie, ce = timeit(eager)
    eager()
    ref = [x.grad.clone()] + [p.grad.clone() for p in params]
    y_ref = conv(x).detach().clone()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3):
            eager()
    torch.cuda.current_stream().wait_stream(s)
    x.grad = None
    for p in params:
        p.grad = None
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        y = conv(x)
        y.backward(g)
    ig, cg = timeit(gr.replay)
    gr.replay()
    torch.cuda.synchronize()
    got = [x.grad] + [p.grad for p in params]
    same = all(torch.equal(torch.view_as_real(a) if a.is_complex() else a, torch.view_as_real(b) if b.is_complex() else b)
               for a, b in zip(got, ref)) and torch.equal(y.detach(), y_ref)
    print(f"{wl}: eager issue {ie:.3f} complete {ce:.3f} ms | graph replay issue {ig:.3f} complete {cg:.3f} ms | "
          f"results bit-identical: {same}", flush=True)
    del gr, x, g, conv, y
    torch.cuda.empty_cache()
