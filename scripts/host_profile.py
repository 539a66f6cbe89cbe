"""Where does the HOST time of a layer step go?  cProfile over N eager steps (dense metric shape, or `tucker`)."""
import cProfile, os, pstats, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neuraloperator_amd import SpectralConv
dev = torch.device("cuda:0")
torch.manual_seed(0)
SP = int(os.environ.get("HP_GRID", "256"))
x = torch.randn(32, 64, SP, SP, device=dev, requires_grad=True)
g = torch.randn(32, 64, SP, SP, device=dev)
kw = dict(factorization="tucker", rank=0.1, implementation="factorized") if "tucker" in sys.argv else {}
conv = SpectralConv(64, 64, (min(64, SP // 2),) * 2, **kw).to(dev)


def step():
    x.grad = None
    for p in conv.parameters():
        p.grad = None
    conv(x).backward(g)


for _ in range(50):
    step()
torch.cuda.synchronize()
N = 200
t0 = time.perf_counter()
for _ in range(N):
    step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"issue {1e3 * (t1 - t0) / N:.3f} ms/step, complete {1e3 * (t2 - t0) / N:.3f} ms/step, cpus {os.cpu_count()}")
pr = cProfile.Profile()
pr.enable()
for _ in range(N):
    step()
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(45)
