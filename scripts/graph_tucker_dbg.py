import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neuraloperator_amd import SpectralConv
from neuraloperator_amd.graph import capture_step
dev = torch.device("cuda:0")
torch.manual_seed(5)
conv = SpectralConv(6, 10, (16, 16), factorization="tucker", rank=0.5, implementation="factorized").to(dev)
x = torch.randn(3, 6, 64, 64, device=dev, requires_grad=True)
g = torch.randn(3, 10, 64, 64, device=dev)
params = [p for p in conv.parameters() if p.requires_grad]
def eager():
    xe = x.detach().clone().requires_grad_(True)
    for p in params: p.grad = None
    ye = conv(xe); ye.backward(g)
    return [ye.detach().clone(), xe.grad.clone()] + [p.grad.clone() for p in params]
a = eager(); b = eager()
f = lambda t: torch.view_as_real(t) if t.is_complex() else t
print("eager vs eager:", [bool(torch.equal(f(u), f(v))) for u, v in zip(a, b)])
step = capture_step(conv, x, g)
y = step.replay().clone()
got = [y, x.grad.clone()] + [p.grad.clone() for p in params]
print("graph vs eager:", [bool(torch.equal(f(u), f(v))) for u, v in zip(got, a)])
print("rel diffs:", [float((f(u) - f(v)).norm() / f(v).norm()) for u, v in zip(got, a)])
print([tuple(p.shape) for p in params])
