"""Phase ablation of the generation-1 matrix-core contraction (k_modegemm_mfma).  The switches are COMPILE-TIME
(-DSC_MG_ABLATE=bits: 1 skip MFMA, 2 skip C stores, 4 skip operand loads), so build the variants first:
    python scripts/build_variants.py g1_nomfma=SC_MG_ABLATE=1 g1_nostore=SC_MG_ABLATE=2 g1_noload=SC_MG_ABLATE=4 \\
        g1_loadsonly=SC_MG_ABLATE=3 g1_mfmaonly=SC_MG_ABLATE=6
    python scripts/gemm_ablate.py neuraloperator_amd/libsc_engine.so neuraloperator_amd/libsc_engine_g1_*.so
(us per launch of the forward contraction at the metric shape, generation 1 forced with SC_GEMM_NO_STREAM;
round-1 numbers: profiles/r01_mfma_gemm_ablation.txt)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neuraloperator_amd import _lib  # noqa: E402

dev = torch.device("cuda:0")
st = torch.cuda.current_stream().cuda_stream
B, C, M = 32, 64, 2112
xh = torch.randn(B, C, M, 2, device=dev)
w = torch.randn(C, C, M, 2, device=dev)
out = torch.empty(B, C, M, 2, device=dev)
junk = torch.empty(600 * 1024 * 1024 // 4, device=dev).normal_()
kw = dict(flags=_lib.SC_GEMM_NO_STREAM, P=B, Q=C, R=C, n_modes=M, a_sp=C * M, a_sr=M, a_sm=1, b_sr=C * M, b_sq=M,
          b_sm=1, c_sp=C * M, c_sq=M, c_sm=1)
for path in sys.argv[1:] or [_lib.DEFAULT_LIB]:
    lib = _lib.ScEngineLib(path)
    fn = lambda: lib.modegemm(xh.data_ptr(), w.data_ptr(), out.data_ptr(), st, **kw)
    res = {}
    for cold in (False, True):
        ts = []
        for _ in range(12):
            if cold:
                junk.sum()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            fn()
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3)
        res["cold" if cold else "warm"] = sorted(ts)[len(ts) // 2]
    print(f"{os.path.basename(path):40s} warm {res['warm']:6.1f} us   cold {res['cold']:6.1f} us", flush=True)
