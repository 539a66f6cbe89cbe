"""Where do the ~30 us between the raw C-ABI step (0.554 ms) and the module step (0.586 ms) go?  The same layer step
in forms that add the module path's ingredients one at a time; ms per step (perf_counter around N steps + sync),
interleaved over ROUNDS rounds."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neuraloperator_amd import SpectralConv, _lib, engine
from neuraloperator_amd.engine import get_plan

dev = torch.device("cuda:0")
B, C, H, W = 32, 64, 256, 256
conv = SpectralConv(C, C, (64, 64)).to(dev)
x = torch.randn(B, C, H, W, device=dev, requires_grad=True)
g = torch.randn(B, C, H, W, device=dev)
lib = _lib.get_lib()
plan = get_plan(dev, [H, W], [64, 33], "forward", 0)
L = lib.layer_desc(B, C, C, [64, 33], [0, 0])
nws = lib.layer_workspace_bytes(plan, L)
w = torch.view_as_real(conv.weight.tensor.detach()).contiguous()
bias = conv.bias.detach().reshape(-1).contiguous()
st = torch.cuda.current_stream().cuda_stream
P = dict(ws=torch.empty(nws, dtype=torch.uint8, device=dev), y=torch.empty_like(x), xh=torch.empty(B, C, 64, 33, 2, device=dev),
         gx=torch.empty_like(x), gw=torch.empty_like(w), gb=torch.empty(C, device=dev))
xd = x.detach()


def run(ws, y, xh, ws2, gx, gw, gb):
    lib.layer_forward(plan, L, xd.data_ptr(), w.data_ptr(), bias.data_ptr(), y.data_ptr(), xh.data_ptr(), ws.data_ptr(), st)
    lib.layer_backward(plan, L, g.data_ptr(), xh.data_ptr(), w.data_ptr(), gx.data_ptr(), gw.data_ptr(), gb.data_ptr(), ws2.data_ptr(), st)


def v0_raw():
    run(P["ws"], P["y"], P["xh"], P["ws"], P["gx"], P["gw"], P["gb"])


def v1_fresh():
    ws = torch.empty(nws, dtype=torch.uint8, device=dev)
    y = torch.empty_like(xd)
    xh = torch.empty(B, C, 64, 33, 2, device=dev)
    ws2 = torch.empty(nws, dtype=torch.uint8, device=dev)
    gx = torch.empty_like(xd)
    gw = torch.empty_like(w)
    gb = torch.empty(C, device=dev)
    run(ws, y, xh, ws2, gx, gw, gb)


held = {}


def v2_fresh_held():            # the gradients of the previous step stay alive until the next one starts (x.grad = None)
    held.clear()
    ws = torch.empty(nws, dtype=torch.uint8, device=dev)
    y = torch.empty_like(xd)
    xh = torch.empty(B, C, 64, 33, 2, device=dev)
    ws2 = torch.empty(nws, dtype=torch.uint8, device=dev)
    gx = torch.empty_like(xd)
    gw = torch.empty_like(w)
    gb = torch.empty(C, device=dev)
    run(ws, y, xh, ws2, gx, gw, gb)
    held.update(gx=gx, gw=gw, gb=gb)


def v3_fn_grad():               # the autograd Function, gradients returned (not accumulated into .grad)
    y = engine.SpectralConvDenseFn.apply(x, conv.weight.tensor, conv.bias, [64, 33], [64, 33], "forward", 0)
    return torch.autograd.grad(y, (x, conv.weight.tensor, conv.bias), g)


def v4_module():
    x.grad = None
    for p in conv.parameters():
        p.grad = None
    y = conv(x)
    y.backward(g)


FNS = [("raw persistent buffers", v0_raw), ("raw, fresh torch.empty per step", v1_fresh),
       ("raw, fresh + gradients held", v2_fresh_held), ("autograd Function + autograd.grad", v3_fn_grad),
       ("module, .backward()", v4_module)]
N, ROUNDS = 40, int(os.environ.get("ROUNDS", 5))
res = {n: [] for n, _ in FNS}
for _ in range(ROUNDS):
    for n, f in FNS:
        for _ in range(3):
            f()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(N):
            f()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        res[n].append(((t2 - t0) / N * 1e3, (t1 - t0) / N * 1e3))
for n, _ in FNS:
    v = sorted(res[n])
    print(f"{n:36s}: total {v[len(v) // 2][0]:.3f} ms/step (min {v[0][0]:.3f})   enqueue {v[len(v) // 2][1]:.3f}")
print("allocator:", {k: v for k, v in torch.cuda.memory_stats().items() if k in ("num_alloc_retries", "reserved_bytes.all.current", "allocated_bytes.all.peak", "num_device_alloc")})
