"""Launch each layer contraction a few times (for rocprofv3 --pmc / --kernel-trace)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neuraloperator_amd import _lib
lib = _lib.get_lib()
dev = torch.device("cuda:0")
B, C, M = 32, 64, 2112
flags = int(sys.argv[1], 0) if len(sys.argv) > 1 else 0
xh = torch.randn(B, C, M, 2, device=dev); gh = torch.randn(B, C, M, 2, device=dev); w = torch.randn(C, C, M, 2, device=dev)
out_s = torch.empty(B, C, M, 2, device=dev); out_w = torch.empty(C, C, M, 2, device=dev)
junk = torch.empty(600 * 1024 * 1024 // 4, device=dev)
st = torch.cuda.current_stream().cuda_stream
for _ in range(5):
    junk.fill_(1.0)
    lib.modegemm(xh.data_ptr(), w.data_ptr(), out_s.data_ptr(), st, flags=flags, P=B, Q=C, R=C, n_modes=M, a_sp=C * M, a_sr=M, a_sm=1, b_sr=C * M, b_sq=M, b_sm=1, c_sp=C * M, c_sq=M, c_sm=1)
    junk.fill_(1.0)
    lib.modegemm(xh.data_ptr(), gh.data_ptr(), out_w.data_ptr(), st, flags=flags, P=C, Q=C, R=B, n_modes=M, a_sp=M, a_sr=C * M, a_sm=1, conj_a=1, b_sr=C * M, b_sq=M, b_sm=1, c_sp=C * M, c_sq=M, c_sm=1)
torch.cuda.synchronize()
