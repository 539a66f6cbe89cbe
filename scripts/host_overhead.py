"""Is the step host-bound?  Enqueue time of N steps vs their GPU time."""
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
    y = conv(x)
    y.backward(g)
MODE = os.environ.get("MODE", "both")        # module | cabi | both  (one path per rocprofv3 run)
N = 50
if MODE in ("module", "both"):
    for _ in range(5):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(N):
        step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"enqueue {1e3 * (t1 - t0) / N:.3f} ms/step   total {1e3 * (t2 - t0) / N:.3f} ms/step")
if MODE == "module":
    sys.exit(0)
# pure C-ABI sequence without autograd / allocations
from neuraloperator_amd import _lib
from neuraloperator_amd.engine import get_plan
lib = _lib.get_lib()
plan = get_plan(dev, [256, 256], [64, 33], "forward", 0)
L = lib.layer_desc(32, 64, 64, [64, 33], [0, 0])
ws = torch.empty(lib.layer_workspace_bytes(plan, L), dtype=torch.uint8, device=dev)
w = torch.view_as_real(conv.weight.tensor.detach()).contiguous()
y = torch.empty_like(x); xh = torch.empty(32, 64, 64, 33, 2, device=dev)
gx = torch.empty_like(x); gw = torch.empty_like(w); gb = torch.empty(64, device=dev)
bias = conv.bias.detach().reshape(-1).contiguous()
st = torch.cuda.current_stream().cuda_stream
def cstep():
    lib.layer_forward(plan, L, x.data_ptr(), w.data_ptr(), bias.data_ptr(), y.data_ptr(), xh.data_ptr(), ws.data_ptr(), st)
    lib.layer_backward(plan, L, g.data_ptr(), xh.data_ptr(), w.data_ptr(), gx.data_ptr(), gw.data_ptr(), gb.data_ptr(), ws.data_ptr(), st)
for _ in range(5):
    cstep()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(N):
    cstep()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"C-ABI only: enqueue {1e3 * (t1 - t0) / N:.3f} ms/step   total {1e3 * (t2 - t0) / N:.3f} ms/step")
