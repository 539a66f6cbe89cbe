"""The activation side of the factorized Tucker contraction at BASELINE configs[2] (B = 32, 64 channels, ranks 36, 2112
modes): the fused one-launch-each-way kernels (csrc/sc_kernels_tkchain.h, round 5) against the nine launches of rounds
3-4, through the C-ABI; agreement with a complex128 einsum on the device, then event-timed calls.
Usage: python scripts/tkchain_time.py [B Ci Co R1 R2 M]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neuraloperator_amd import _lib  # noqa: E402

dev = torch.device("cuda:0")
dims = tuple(int(v) for v in sys.argv[1:7]) if len(sys.argv) >= 7 else (32, 64, 64, 36, 36, 2112)
B, Ci, Co, R1, R2, M = dims
lib = _lib.get_lib()
st = torch.cuda.current_stream().cuda_stream
torch.manual_seed(0)
rnd = lambda *sh: torch.randn(*sh, dtype=torch.complex64, device=dev)
xhat, u_in, t3, u_out, gy = rnd(B, Ci, M), rnd(Ci, R1), rnd(R1, R2, M), rnd(Co, R2), rnd(B, Co, M)
new = lambda *sh: torch.full(sh, float("nan"), dtype=torch.complex64, device=dev)
p = lambda t: 0 if t is None else t.data_ptr()
rel = lambda a, b: float((a.to(torch.complex128) - b).norm() / b.norm())

c = lambda v: v.to(torch.complex128)
Z = torch.einsum("bim,if->bfm", c(xhat), c(u_in))
T = torch.einsum("bfm,fgm->bgm", Z, c(t3))
Y = torch.einsum("bgm,og->bom", T, c(u_out))
gT = torch.einsum("bom,og->bgm", c(gy), c(u_out).conj())
gUo = torch.einsum("bgm,bom->og", T.conj(), c(gy))
gZ = torch.einsum("bgm,fgm->bfm", gT, c(t3).conj())
gT3 = torch.einsum("bfm,bgm->fgm", Z.conj(), gT)
gX = torch.einsum("bfm,if->bim", gZ, c(u_in).conj())
gUi = torch.einsum("bim,bfm->if", c(xhat).conj(), gZ)


def timed(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(3):
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / n * 1e3)
    return best


out = {}
for name in ("nine launches", "fused"):
    fused = name == "fused"
    if fused and not lib.tucker_chain_fused_supported(dims):
        print("fused: shape not supported")
        continue
    z, t, yhat, t3m = new(B, R1, M), new(B, R2, M), new(B, Co, M), new(M, R1, R2)
    gx, gui, gt3, guo = new(B, Ci, M), new(Ci, R1), new(R1, R2, M), new(Co, R2)
    if fused:
        nb = lib.tucker_chain_backward_fused_workspace_bytes(dims)
        ws = torch.empty(nb, dtype=torch.uint8, device=dev)
        fwd = lambda: lib.tucker_chain_forward_fused(dims, p(xhat), p(u_in), p(t3), p(u_out), p(t3m), p(z), p(t), p(yhat), st)
        bwd = lambda: lib.tucker_chain_backward_fused(dims, p(xhat), p(u_in), p(t3m), p(u_out), p(z), p(t), p(gy), p(gx), p(gui),
                                                      p(gt3), p(guo), ws.data_ptr(), nb, st)
    else:
        nb = lib.tucker_chain_workspace_bytes(dims)
        ws = torch.empty(nb, dtype=torch.uint8, device=dev)
        fwd = lambda: lib.tucker_chain_forward(dims, p(xhat), p(u_in), p(t3), p(u_out), p(z), p(t), p(yhat), st)
        bwd = lambda: lib.tucker_chain_backward(dims, p(xhat), p(u_in), p(t3), p(u_out), p(z), p(t), p(gy), p(gx), p(gui), p(gt3),
                                                p(guo), ws.data_ptr(), nb, st)
    fwd()
    bwd()
    torch.cuda.synchronize()
    errs = dict(z=rel(z, Z), t=rel(t, T), yhat=rel(yhat, Y), gx=rel(gx, gX), gt3=rel(gt3, gT3), gu_in=rel(gui, gUi), gu_out=rel(guo, gUo))
    tf, tb = timed(fwd), timed(bwd)
    out[name] = (tf, tb)
    print(f"{name:>14}: forward {tf:7.1f} us  backward {tb:7.1f} us  sum {tf + tb:7.1f} us | rel-L2 vs complex128: " +
          " ".join(f"{k} {v:.1e}" for k, v in errs.items()))
    assert all(v < 3e-6 for v in errs.values()), errs
if len(out) == 2:
    a, b = out["nine launches"], out["fused"]
    print(f"dims {dims}: fused / nine launches = {b[0] / a[0]:.2f} forward, {b[1] / a[1]:.2f} backward")
