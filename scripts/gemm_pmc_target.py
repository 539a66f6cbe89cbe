"""rocprofv3 --pmc target: the forward contraction of the metric shape, 6 launches, each after a clean
(read-only) eviction of the caches.  python scripts/gemm_pmc_target.py [tiled]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neuraloperator_amd import _lib  # noqa: E402

lib = _lib.get_lib()
dev = torch.device("cuda:0")
st = torch.cuda.current_stream().cuda_stream
B, C, M = 32, 64, 2112
tiled = len(sys.argv) > 1 and sys.argv[1] == "tiled"
x = torch.randn(B * C * M * 2, device=dev)
w = torch.randn(C * C * M * 2, device=dev)
y = torch.empty(B * C * M * 2, device=dev)
junk = torch.empty(600 * 1024 * 1024 // 4, device=dev).normal_()
kw = dict(P=B, Q=C, R=C, n_modes=M, a_sm=1, b_sm=1, c_sm=1)
if tiled:
    kw.update(a_sg=B * C * 16, a_sp=C * 16, a_sr=16, b_sg=C * C * 16, b_sr=C * 16, b_sq=16, c_sg=B * C * 16, c_sp=C * 16, c_sq=16)
else:
    kw.update(a_sp=C * M, a_sr=M, b_sr=C * M, b_sq=M, c_sp=C * M, c_sq=M)
for _ in range(6):
    s = junk.sum()
    lib.modegemm(x.data_ptr(), w.data_ptr(), y.data_ptr(), st, **kw)
torch.cuda.synchronize()
print("done", float(s) * 0)
