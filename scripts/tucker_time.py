"""Time the two Tucker mode-factor launches alone (configs[2]: 1296 (f, g) pairs, ranks (36, 19), kept 64 x 33)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neuraloperator_amd import _lib
lib = _lib.get_lib()
dev = torch.device("cuda:0")
fg, rx, ry, mx, my = 1296, 36, 19, 64, 33
c = lambda *s: torch.randn(*s, dtype=torch.complex64, device=dev)
core, ux, uy, gt = c(fg, rx, ry), c(mx, rx), c(my, ry), c(fg, mx, my)
t, gc, ga, gb = c(fg, mx, my), c(fg, rx, ry), c(mx, rx), c(my, ry)
ws = torch.empty(lib.tucker_modes_workspace_bytes(fg, rx, ry, mx, my), dtype=torch.uint8, device=dev)
p = lambda z: z.data_ptr()
st = torch.cuda.current_stream().cuda_stream
def fwd(): lib.tucker_modes_forward(fg, rx, ry, mx, my, p(core), p(ux), p(uy), p(t), st)
def bwd(): lib.tucker_modes_backward(fg, rx, ry, mx, my, p(core), p(ux), p(uy), p(gt), p(gc), p(ga), p(gb), p(ws), st)
for name, fn in (("fwd", fwd), ("bwd", bwd)):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): fn()
    e1.record(); torch.cuda.synchronize()
    print(f"{os.environ.get('TAG', '')} tucker modes {name}: {e0.elapsed_time(e1) / 50 * 1e3:.1f} us")
