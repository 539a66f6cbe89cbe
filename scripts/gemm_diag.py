"""Diagnostics of the streamed contraction (k_modegemm_dma), one box:
  (1) time vs number of mode groups (same per-workgroup work): bandwidth-bound kernels scale with the grid,
      latency-bound ones do not;
  (2) operand layouts: plain reference layout vs mode-group-major ("tiled") A / C vs everything tiled.
us per launch: warm (back to back), cold-r (after READING 600 MB: clean eviction), cold-w (after a 600 MB fill)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neuraloperator_amd import _lib  # noqa: E402

lib = _lib.ScEngineLib(sys.argv[1] if len(sys.argv) > 1 else _lib.DEFAULT_LIB)
dev = torch.device("cuda:0")
st = torch.cuda.current_stream().cuda_stream
junk = torch.empty(600 * 1024 * 1024 // 4, device=dev).normal_()
sink = torch.zeros(1, device=dev)


def timed(fn, mode, n=10):
    tot = []
    for _ in range(n):
        if mode == "cold-w":
            junk.fill_(1.0)
        elif mode == "cold-r":
            sink.add_(junk.sum())
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        tot.append(e0.elapsed_time(e1) * 1e3)
    tot.sort()
    return tot[len(tot) // 2]


def run(tag, B, C, M, layout, flags=0):
    """fwd contraction y[b,o,m] = sum_i x[b,i,m] w[i,o,m]; layout: which operands are mode-group-major"""
    G = M // 16
    x = torch.randn(B * C * M * 2, device=dev)
    w = torch.randn(C * C * M * 2, device=dev)
    y = torch.empty(B * C * M * 2, device=dev)
    kw = dict(P=B, Q=C, R=C, n_modes=M, a_sm=1, b_sm=1, c_sm=1, flags=flags)
    kw.update(dict(a_sg=B * C * 16, a_sp=C * 16, a_sr=16) if "A" in layout else dict(a_sp=C * M, a_sr=M))
    kw.update(dict(b_sg=C * C * 16, b_sr=C * 16, b_sq=16) if "B" in layout else dict(b_sr=C * M, b_sq=M))
    kw.update(dict(c_sg=B * C * 16, c_sp=C * 16, c_sq=16) if "C" in layout else dict(c_sp=C * M, c_sq=M))
    fn = lambda: lib.modegemm(x.data_ptr(), w.data_ptr(), y.data_ptr(), st, **kw)
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    nbytes = (x.numel() + w.numel() + y.numel()) * 4
    t = {m: timed(fn, m) for m in ("warm", "cold-r", "cold-w")}
    print(f"{tag:34s} path {lib.modegemm_path(**kw)} groups {G:5d} {nbytes / 1e6:7.1f} MB | " +
          " | ".join(f"{m} {v:6.1f} us {nbytes / v / 1e6:5.2f} TB/s" for m, v in t.items()), flush=True)


for M in (1024, 2048, 2112, 3072, 4096, 8448):
    run(f"plain M={M}", 32, 64, M, "")
for lay in ("", "AC", "B", "ABC"):
    run(f"M=2112 tiled[{lay or '-'}]", 32, 64, 2112, lay)
run("M=2112 generation 1", 32, 64, 2112, "", flags=_lib.SC_GEMM_NO_STREAM)
