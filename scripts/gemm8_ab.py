"""A-B of the layer's three contractions between kernels / engine builds on one box, interleaved:
    python scripts/gemm8_ab.py [lib.so ...]
For every library: the streamed kernel (k_modegemm_s8, flags 0) and generation 1 (SC_GEMM_NO_STREAM) at the layer's
own strides; us per launch warm (back to back) and cold (after READING 600 MB: caches evicted, nothing dirty)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neuraloperator_amd import _lib  # noqa: E402

paths = sys.argv[1:] or [_lib.DEFAULT_LIB]
libs = [(os.path.basename(p).replace("libsc_engine", "").replace(".so", "") or "prod", _lib.ScEngineLib(p)) for p in paths]
dev = torch.device("cuda:0")
st = torch.cuda.current_stream().cuda_stream
junk = torch.empty(600 * 1024 * 1024 // 4, device=dev).normal_()
sink = torch.zeros(1, device=dev)


def timed(fn, cold, n=12):
    tot = []
    for _ in range(n):
        if cold:
            sink.add_(junk.sum())        # clean eviction: a 600 MB fill would leave dirty lines the timed kernel pays for
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        tot.append(e0.elapsed_time(e1) * 1e3)
    tot.sort()
    return tot[len(tot) // 2], tot[0]


def cases(B, C, M):
    return [
        ("fwd", dict(P=B, Q=C, R=C, n_modes=M, a_sp=C * M, a_sr=M, a_sm=1, b_sr=C * M, b_sq=M, b_sm=1,
                     c_sp=C * M, c_sq=M, c_sm=1), (B, C), (C, C), (B, C)),
        ("gW ", dict(P=C, Q=C, R=B, n_modes=M, a_sp=M, a_sr=C * M, a_sm=1, conj_a=1, b_sr=C * M, b_sq=M, b_sm=1,
                     c_sp=C * M, c_sq=M, c_sm=1, flags=_lib.SC_GEMM_STREAM_C), (B, C), (B, C), (C, C)),
        ("gX ", dict(P=B, Q=C, R=C, n_modes=M, a_sp=C * M, a_sr=M, a_sm=1, b_sr=M, b_sq=C * M, b_sm=1, conj_b=1,
                     c_sp=C * M, c_sq=M, c_sm=1), (B, C), (C, C), (B, C)),
    ]


for tag, B, C, M in [("fno2d_256 B32 C64 M2112", 32, 64, 2112), ("fno2d_1024 B4 C128 M33024", 4, 128, 33024),
                     ("B32 C128 M2112", 32, 128, 2112)]:
    for name, kw, sa, sb, sc in cases(B, C, M):
        a = torch.randn(*sa, M, 2, device=dev)
        b = torch.randn(*sb, M, 2, device=dev)
        c = torch.empty(*sc, M, 2, device=dev)
        nbytes = (a.numel() + b.numel() + c.numel()) * 4
        row = [f"{tag} {name} {nbytes / 1e6:7.1f} MB"]
        for lname, lib in libs:
            for vname, extra in (("s8", 0), ("g1", _lib.SC_GEMM_NO_STREAM)):
                k = dict(kw)
                k["flags"] = k.get("flags", 0) | extra
                path = lib.modegemm_path(**k)
                fn = lambda: lib.modegemm(a.data_ptr(), b.data_ptr(), c.data_ptr(), st, **k)
                for _ in range(3):
                    fn()
                torch.cuda.synchronize()
                (wm, wmin), (cm, cmin) = timed(fn, False), timed(fn, True)
                row.append(f"{lname}/{vname}[{path}] warm {wm:6.1f} cold {cm:6.1f} (min {cmin:6.1f}) = {nbytes / cm / 1e6:5.2f} TB/s")
                if lname != libs[0][0] and vname == "g1":
                    row.pop()          # generation 1 is the same code in every build: once is enough
        print(" | ".join(row), flush=True)
