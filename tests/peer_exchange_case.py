"""One rank of tests/test_gpu_peer_exchange.py: RANK WORLD PORT.  Every rank uses cuda:0 (one-GPU box), the process group
is gloo; the peer-store exchange moves the data through HIP-IPC-mapped windows of the SAME device."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
rank, world, port = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=port, RANK=str(rank), WORLD_SIZE=str(world), SC_MPU_A2A="peer")
torch.cuda.set_device(0)
dev = torch.device("cuda:0")
dist.init_process_group("gloo", rank=rank, world_size=world)
from neuraloperator_amd.mpu import ModeParallelSpectralConv, comm, peer_exchange  # noqa: E402

comm.init(model_parallel_size=world)
grp = comm.get_model_parallel_group()


def a2a_ref(send):
    hs, hw = send.cpu(), torch.empty_like(send, device="cpu")
    dist.all_to_all_single(hw, hs, group=grp)
    return hw.to(dev)


# ---- 1. the exchange itself: sizes below / above the first window (regrow), many epochs, a side stream
px = peer_exchange.get(grp, 1 << 16)
assert px is not None, peer_exchange.LAST_REASON
g = torch.Generator(device=dev).manual_seed(77 + rank)
side = torch.cuda.Stream()
for it, n in enumerate([64, 4096, 139264, 64, 1 << 19, 1028]):
    send = torch.randn(world, n, device=dev, generator=g)
    want = a2a_ref(send)
    px = peer_exchange.get(grp, send.numel() * 4)
    got = torch.full_like(send, float("nan"))
    with torch.cuda.stream(side if it % 2 else torch.cuda.current_stream()):
        px.all_to_all(send, got, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert torch.equal(got, want), (rank, n)
# ---- 2. three exchanges recorded into a hipGraph, replayed: every replay signals a fresh epoch
send = torch.randn(world, 2048, device=dev, generator=g)
bufs = [torch.full_like(send, float("nan")) for _ in range(3)]
want = a2a_ref(send)
want2 = a2a_ref(want)
with torch.cuda.stream(side):
    px.all_to_all(send, bufs[0], side.cuda_stream)            # warm-up outside the capture
torch.cuda.synchronize()
dist.barrier()
gr = torch.cuda.CUDAGraph()
with torch.cuda.graph(gr):
    st = torch.cuda.current_stream().cuda_stream
    px.all_to_all(send, bufs[0], st)
    px.all_to_all(bufs[0], bufs[1], st)
    px.all_to_all(send, bufs[2], st)
for rep in range(3):
    for b in bufs:
        b.fill_(float("nan"))
    gr.replay()
    torch.cuda.synchronize()
    assert torch.equal(bufs[0], want) and torch.equal(bufs[1], want2) and torch.equal(bufs[2], want), (rank, rep)
# ---- 3. the mode-parallel layer: peer-store exchanges against the torch path, same bits
torch.manual_seed(5)
conv = ModeParallelSpectralConv(4, 4, (2 * world, 8, 8)).to(dev)
conv.sync_replicated_parameters()
x = torch.randn(1, 4, 4 * world, 16, 16, device=dev, generator=g)
gy = torch.randn(1, 4, 4 * world, 16, 16, device=dev, generator=g)


def step():
    xx = x.clone().requires_grad_(True)
    for q in conv.parameters():
        q.grad = None
    y = conv(xx)
    y.backward(gy)
    torch.cuda.synchronize()
    return [y.detach().clone(), xx.grad.clone(), torch.view_as_real(conv.weight.grad).clone()]


from neuraloperator_amd.mpu import mappings  # noqa: E402
a = step()
assert peer_exchange.active()
os.environ["SC_MPU_A2A"] = "torch"
b = step()
for u, v in zip(a, b):
    assert torch.equal(u, v), rank
dist.barrier()
print(f"[peer-exchange] rank {rank} of {world}: ok ({px.count} exchanges)", flush=True)
comm.cleanup()
