"""Time the MFMA DFT passes one at a time through the C-ABI (1-D plans = last-axis passes only,
2-D plans add one axis pass).  Usage: python scripts/mdft_time.py [lib.so ...]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neuraloperator_amd import _lib
libs = sys.argv[1:] or [_lib.DEFAULT_LIB]
# MDFT_CASES="N,J,lines;N,J,lines": other line lengths (e.g. 421,17,215552 = one 16 x 32 x 421^2 tensor)
CASES = [tuple(int(v) for v in c.split(",")) for c in os.environ.get("MDFT_CASES", "").split(";") if c] or \
    [(128, 17, 8 * 32 * 128 * 128), (1024, 129, 4 * 128 * 1024)]
dev = torch.device("cuda:0")
st = torch.cuda.current_stream().cuda_stream


def timed(fn, n=6):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for path in libs:
    lib = _lib.ScEngineLib(path)
    row = [os.path.basename(path)]
    for (N, J, lines) in CASES:
        plan = lib.plan_create([N], [J])
        x = torch.randn(lines, N, device=dev); y = torch.empty_like(x)
        xh = torch.randn(lines, J, 2, device=dev)
        ws = torch.empty(max(lib.plan_workspace_bytes(plan, lines), 256), dtype=torch.uint8, device=dev)
        tf = timed(lambda: lib.transform_forward(plan, 0, x.data_ptr(), xh.data_ptr(), lines, ws.data_ptr(), st))
        ti = timed(lambda: lib.transform_inverse(plan, 0, xh.data_ptr(), 0, 1, y.data_ptr(), lines, ws.data_ptr(), st))
        gb = lines * N * 4 / 1e9
        row.append(f"N={N} J={J}: r2c {tf:7.1f} us ({gb / tf * 1e3:5.2f} TB/s)  c2r {ti:7.1f} us ({gb / ti * 1e3:5.2f} TB/s)")
        lib.plan_destroy(plan)
        del x, y, xh, ws
    print(" | ".join(row), flush=True)
