"""Per-rank step of the mode-parallel layer on ONE GPU (a one-rank RCCL group): where the host time goes (cProfile),
and whether the whole step -- engine launches AND the RCCL exchanges -- records into one hipGraph.
    MS_SHAPE = "B,C,spatial...,modes..."  (default: configs[3]'s per-rank share at 8 GPUs: 1,32,128,128,128,32,32,32)
    MS_CHUNKS, MS_CHUNK_DIM"""
import cProfile
import functools
import os
import pstats
import sys
import time

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neuraloperator_amd.mpu import ModeParallelSpectralConv, comm  # noqa: E402

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
print = functools.partial(print, flush=True)
dev = torch.device("cuda:0")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
comm.init(model_parallel_size=1)
shape = [int(v) for v in os.environ.get("MS_SHAPE", "1,32,128,128,128,32,32,32").split(",")]
B, C = shape[:2]
nd = (len(shape) - 2) // 2
spatial, modes = shape[2:2 + nd], shape[2 + nd:]
kw = {}
if os.environ.get("MS_CHUNKS"):
    kw["comm_chunks"] = int(os.environ["MS_CHUNKS"])
if os.environ.get("MS_CHUNK_DIM"):
    kw["chunk_dim"] = os.environ["MS_CHUNK_DIM"]
conv = ModeParallelSpectralConv(C, C, tuple(modes), **kw).to(dev)
x = torch.randn(B, C, *spatial, device=dev, requires_grad=True)
g = torch.randn(B, C, *spatial, device=dev)


def step():
    x.grad = None
    for q in conv.parameters():
        q.grad = None
    conv(x).backward(g)


def timed(fn, n=50):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    return (t1 - t0) / n * 1e3, (t2 - t0) / n * 1e3


issue, total = timed(step)
from neuraloperator_amd.mpu import rccl_native  # noqa: E402
print(f"eager: host issue {issue:.3f} ms/step, wall {total:.3f} ms/step  ({kw}; exchanges: "
      f"{'native RCCL' if rccl_native.get(conv._group()) is not None else 'torch.distributed (' + rccl_native.LAST_REASON + ')'})")
pr = cProfile.Profile()
pr.enable()
for _ in range(50):
    step()
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats(os.environ.get("MS_SORT", "cumulative")).print_stats(int(os.environ.get("MS_TOP", 32)))
sys.stdout.flush()
if os.environ.get("MS_GRAPH") != "1":
    comm.cleanup()
    sys.exit(0)

# the whole step as ONE hipGraph, collectives included
try:
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            step()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    x.grad = None
    for q in conv.parameters():
        q.grad = None
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        y = conv(x)
        y.backward(g)
    torch.cuda.synchronize()
    ref_y = y.detach().clone()
    ref_gx = x.grad.clone()
    gr.replay()
    torch.cuda.synchronize()
    same = torch.equal(ref_y, y.detach()) and torch.equal(ref_gx, x.grad)
    issue, total = timed(gr.replay)
    print(f"hipGraph replay (RCCL exchanges captured): issue {issue:.3f} ms, wall {total:.3f} ms/step, replay == capture run: {same}")
    xe = x.detach().clone().requires_grad_(True)
    ye = conv(xe)
    ye.backward(g)
    torch.cuda.synchronize()
    print("replay vs eager: y", float((ye.detach() - y.detach()).abs().max()), "gx", float((xe.grad - x.grad).abs().max()))
except Exception as e:
    print("graph capture failed:", type(e).__name__, str(e)[:300])
comm.cleanup()
