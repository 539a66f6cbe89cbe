"""Time the plane-form passes (last two axes in one launch) through the C-ABI: 2-D plans of 128 x 128 planes,
32768 of them (= the 128^3, B 8 x C 32 tensor).  Usage: python scripts/plane_time.py [lib.so ...]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neuraloperator_amd import _lib
libs = sys.argv[1:] or [_lib.DEFAULT_LIB]
dev = torch.device("cuda:0")
st = torch.cuda.current_stream().cuda_stream
planes, N1, N2, K1, J = 8 * 32 * 128, 128, 128, 32, 17
x = torch.randn(planes, N1, N2, device=dev); y = torch.empty_like(x)
z = torch.randn(planes, K1, J, 2, device=dev)


def timed(fn, n=20):
    for _ in range(30):
        fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for path in libs * (3 if len(libs) > 1 else 1):      # interleaved repetitions: the clocks of a fresh box settle over the first passes
    lib = _lib.ScEngineLib(path)
    plan = lib.plan_create([N1, N2], [K1, J])
    ws = torch.empty(max(lib.plan_workspace_bytes(plan, planes), 256), dtype=torch.uint8, device=dev)
    tf = timed(lambda: lib.transform_forward(plan, 0, x.data_ptr(), z.data_ptr(), planes, ws.data_ptr(), st))
    ti = timed(lambda: lib.transform_inverse(plan, 0, z.data_ptr(), 0, 1, y.data_ptr(), planes, ws.data_ptr(), st))
    gb = planes * N1 * N2 * 4 / 1e9
    print(f"{os.path.basename(path)}: {lib.plan_kernel_name(plan, 0)} {tf:7.1f} us ({gb / tf * 1e3:5.2f} TB/s)   "
          f"{lib.plan_kernel_name(plan, 1)} {ti:7.1f} us ({gb / ti * 1e3:5.2f} TB/s)", flush=True)
    lib.plan_destroy(plan)
