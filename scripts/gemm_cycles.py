import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neuraloperator_amd import _lib
lib = _lib.get_lib()
dev = torch.device("cuda:0")
B, C, M = 32, 64, 2112
xh = torch.randn(B, C, M, 2, device=dev); w = torch.randn(C, C, M, 2, device=dev)
out_s = torch.zeros(B, C, M, 2, device=dev)
st = torch.cuda.current_stream().cuda_stream
for _ in range(3):
    lib.modegemm(xh.data_ptr(), w.data_ptr(), out_s.data_ptr(), st, flags=14 << 24, P=B, Q=C, R=C, n_modes=M, a_sp=C * M, a_sr=M, a_sm=1, b_sr=C * M, b_sq=M, b_sm=1, c_sp=C * M, c_sq=M, c_sm=1)
torch.cuda.synchronize()
v = out_s.view(-1)[:32].view(torch.int64).cpu().tolist()
for i in range(8):
    cyc, wall = v[2 * i], v[2 * i + 1]
    print(f"block {i}: loop {cyc} clock64 ticks = {cyc / (8 * 72):.1f} per MFMA; wall {wall / 100:.1f} us -> {cyc / (wall / 100e6) / 1e9:.2f} GHz")
