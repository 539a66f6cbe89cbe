"""Large-grid transforms (BASELINE configs[4]: 1024 x 1024, modes 256, B x C = 512): the two-pass factorised route
against the size-agnostic direct-DFT route of the same library -- agreement on the GPU, then event-timed launches.
Usage: python scripts/f2p_time.py [lib.so ...]   (extra libraries = chunk-size / occupancy variants)"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neuraloperator_amd import _lib

dev = torch.device("cuda:0")
N0 = N1 = int(os.environ.get("F2P_N", 1024))
K0, J = int(os.environ.get("F2P_K0", 256)), int(os.environ.get("F2P_J", 129))
NIMG = int(os.environ.get("F2P_IMAGES", 512))
st = torch.cuda.current_stream().cuda_stream


def run(lib, flags, x, yh, bias, n_img, reps):
    plan = lib.plan_create([N0, N1], [K0, J], flags=flags)
    ws = torch.empty(max(lib.plan_workspace_bytes(plan, n_img), 256), dtype=torch.uint8, device=dev)
    xh = torch.empty(n_img, K0, J, 2, device=dev)
    y = torch.empty(n_img, N0, N1, device=dev)
    f = lambda m: lib.transform_forward(plan, m, x.data_ptr(), xh.data_ptr(), n_img, ws.data_ptr(), st)
    i = lambda m: lib.transform_inverse(plan, m, yh.data_ptr(), bias.data_ptr() if m == 0 else 0, bias.numel(), y.data_ptr(),
                                        n_img, ws.data_ptr(), st)
    out = {}
    for m in (0, 1):
        f(m); out[f"fwd{m}"] = xh.clone()
        i(m); out[f"inv{m}"] = y.clone()
    torch.cuda.synchronize()
    t = {}
    if reps:
        for name, fn in (("fwd", lambda: f(0)), ("inv", lambda: i(0)), ("adj_c2r", lambda: f(1)), ("adj_r2c", lambda: i(1))):
            fn(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                fn()
            e1.record(); torch.cuda.synchronize()
            t[name] = e0.elapsed_time(e1) / reps
    name = lib.plan_kernel_name(plan, 0)
    lib.plan_destroy(plan)
    return out, t, name


libs = sys.argv[1:] or [_lib.DEFAULT_LIB]
torch.manual_seed(0)
x = torch.randn(NIMG, N0, N1, device=dev)
yh = torch.randn(NIMG, K0, J, 2, device=dev)
bias = torch.randn(128, device=dev)
alg = (NIMG * N0 * N1 * 4 + NIMG * K0 * J * 8) / 1e9
base = None
for path in libs:
    lib = _lib.ScEngineLib(path)
    if base is None:
        nchk = min(NIMG, 6)
        a, _, na = run(lib, 0, x[:nchk], yh[:nchk], bias, nchk, 0)
        b, _, nb = run(lib, _lib.SC_PLAN_FORCE_GENERIC, x[:nchk], yh[:nchk], bias, nchk, 0)
        for k in a:
            err = ((a[k] - b[k]).norm() / b[k].norm()).item()
            print(f"agreement {na} vs {nb} {k}: rel-L2 {err:.2e}")
        _, tg, _ = run(lib, _lib.SC_PLAN_FORCE_GENERIC, x, yh, bias, NIMG, 3)
        print("direct-DFT route  :", "  ".join(f"{k} {v:.3f} ms ({alg / v * 1e3:.0f} GB/s)" for k, v in tg.items()))
        base = True
    _, t, nm = run(lib, 0, x, yh, bias, NIMG, 10)
    print(f"{os.path.basename(path)} [{nm}]:", "  ".join(f"{k} {v:.3f} ms ({alg / v * 1e3:.0f} GB/s)" for k, v in t.items()))
