"""Fused AdamW step (sc_adamw_step) vs the reference's chain of elementwise ATen ops on the 69 MB complex
spectral weight of the metric layer (64 x 64 x 64 x 33 complex64).  us per step."""
import math, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neuraloperator_amd import AdamW
dev = torch.device("cuda:0")


def ref_step(p, grad, state, lr=1e-3, b1=0.9, b2=0.999, eps=1e-6, wd=0.0):
    # the arithmetic of neuralop/training/adamw.py:155-200 (non-GaLore branch), op for op
    m, v = state["m"], state["v"]
    state["step"] += 1
    m.mul_(b1).add_(grad, alpha=1.0 - b1)
    v.mul_(b2).addcmul_(grad, grad.conj(), value=1.0 - b2)
    denom = v.sqrt().add_(eps)
    ss = lr * math.sqrt(1.0 - b2 ** state["step"]) / (1.0 - b1 ** state["step"])
    p.add_(m / denom, alpha=-ss)
    if wd > 0:
        p.add_(p, alpha=-lr * wd)


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


w = torch.nn.Parameter(torch.randn(64, 64, 64, 33, dtype=torch.cfloat, device=dev))
w.grad = torch.randn_like(w)
opt = AdamW([w], lr=1e-3, weight_decay=1e-4)
t_fused = timed(opt.step)
p2 = torch.randn(64, 64, 64, 33, dtype=torch.cfloat, device=dev)
g2 = torch.randn_like(p2)
st = dict(m=torch.zeros_like(p2), v=torch.zeros_like(p2), step=0)
with torch.no_grad():
    t_ref = timed(lambda: ref_step(p2, g2, st, wd=1e-4))
mb = w.numel() * 8 / 1e6
print(f"weight {mb:.1f} MB: fused {t_fused:.1f} us ({7 * mb / t_fused:.2f} TB/s over 7 arrays)   "
      f"elementwise chain {t_ref:.1f} us   ({t_ref / t_fused:.1f}x)")
