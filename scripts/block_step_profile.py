"""Kernel breakdown of one fused FNO block step (bench.py's extra.fno_block, fused path only): run under
rocprofv3 --kernel-trace --stats."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
dev = torch.device("cuda:0")
B, C, spatial, n_modes = bench.WORKLOADS["fno2d_256_m64_c64_b32"]
print(bench.block_extra(B, C, spatial, n_modes, dev))
