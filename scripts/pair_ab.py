"""A-B of the backward pass's contraction stage between engine builds on one box, interleaved:
    python scripts/pair_ab.py [lib.so ...]
For every library, at the metric shape (B=32, C=64, 256 x 256, modes 64 x 64 -> 2112 kept modes) or SHAPE=...:
  * `seq`  : k_bias_grad + the weight-gradient launch + the spectrum-gradient launch (sc_bias_grad, sc_modegemm x 2)
  * `pair` : sc_modegemm_pair (one launch of k_modegemm_dma_bwd when the build has it)
  * `fwd`  : the forward contraction alone (sc_modegemm)
  * `step` : the whole layer step through sc_layer_forward + sc_layer_backward (what bench.py times)
us per repetition, median and minimum of ROUNDS rounds; plus a bit comparison of `pair` against `seq`."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neuraloperator_amd import _lib  # noqa: E402

paths = sys.argv[1:] or [_lib.DEFAULT_LIB]
libs = [(os.path.basename(p).replace("libsc_engine", "").replace(".so", "").strip("_") or "prod", _lib.ScEngineLib(p))
        for p in paths]
dev = torch.device("cuda:0")
st = torch.cuda.current_stream().cuda_stream
ROUNDS = int(os.environ.get("ROUNDS", 7))
REPS = int(os.environ.get("REPS", 20))
# SHAPE="B,C,spatial...,kept..." (default: the metric shape; FNO3d 128^3: SHAPE=8,32,128,128,128,32,32,17)
shape = [int(v) for v in os.environ.get("SHAPE", "32,64,256,256,64,33").split(",")]
B, C = shape[:2]
nd = (len(shape) - 2) // 2
SPATIAL, KEPT = shape[2:2 + nd], shape[2 + nd:]
M = 1
for k in KEPT:
    M *= k
torch.manual_seed(0)
x = torch.randn(B, C, *SPATIAL, device=dev)
g = torch.randn(B, C, *SPATIAL, device=dev)
w = torch.randn(C, C, *KEPT, 2, device=dev)
bias = torch.randn(C, device=dev)
y = torch.empty_like(x)
gx = torch.empty_like(x)
gw = torch.empty_like(w)
gb = torch.empty(C, device=dev)
xhat = torch.randn(B, C, M, 2, device=dev)
ghat = torch.randn(B, C, M, 2, device=dev)
gxhat = torch.empty(B, C, M, 2, device=dev)

kw_w = dict(P=C, Q=C, R=B, n_modes=M, a_sp=M, a_sr=C * M, a_sm=1, conj_a=1, b_sr=C * M, b_sq=M, b_sm=1,
            c_sp=C * M, c_sq=M, c_sm=1, flags=_lib.SC_GEMM_STREAM_C)
kw_x = dict(P=B, Q=C, R=C, n_modes=M, a_sp=C * M, a_sr=M, a_sm=1, b_sr=M, b_sq=C * M, b_sm=1, conj_b=1,
            c_sp=C * M, c_sq=M, c_sm=1)


kw_f = dict(P=B, Q=C, R=C, n_modes=M, a_sp=C * M, a_sr=M, a_sm=1, b_sr=C * M, b_sq=M, b_sm=1,
            c_sp=C * M, c_sq=M, c_sm=1)


def timed(fn, reps=REPS):
    e0 = torch.cuda.Event(enable_timing=True)
    e1 = torch.cuda.Event(enable_timing=True)
    fn()
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


ctx = {}
for name, lib in libs:
    plan = lib.plan_create(SPATIAL, KEPT, fft_norm="forward", flags=0)
    L = lib.layer_desc(B, C, C, KEPT, [0] * nd)
    ws = torch.empty(lib.layer_workspace_bytes(plan, L), dtype=torch.uint8, device=dev)
    ctx[name] = (plan, L, ws)


def make(name, lib):
    plan, L, ws = ctx[name]

    def seq():
        lib.bias_grad(plan, ghat.data_ptr(), B, C, gb.data_ptr(), st)
        lib.modegemm(xhat.data_ptr(), ghat.data_ptr(), gw.data_ptr(), st, **kw_w)
        lib.modegemm(ghat.data_ptr(), w.data_ptr(), gxhat.data_ptr(), st, **kw_x)

    def pair():
        lib.bias_grad(plan, ghat.data_ptr(), B, C, gb.data_ptr(), st)
        lib.modegemm_pair(kw_w, xhat.data_ptr(), ghat.data_ptr(), gw.data_ptr(),
                          kw_x, ghat.data_ptr(), w.data_ptr(), gxhat.data_ptr(), st)

    def step():
        lib.layer_forward(plan, L, x.data_ptr(), w.data_ptr(), bias.data_ptr(), y.data_ptr(), xhat.data_ptr(),
                          ws.data_ptr(), st)
        lib.layer_backward(plan, L, g.data_ptr(), xhat.data_ptr(), w.data_ptr(), gx.data_ptr(), gw.data_ptr(),
                           gb.data_ptr(), ws.data_ptr(), st)

    def bwd():
        lib.layer_backward(plan, L, g.data_ptr(), xhat.data_ptr(), w.data_ptr(), gx.data_ptr(), gw.data_ptr(),
                           gb.data_ptr(), ws.data_ptr(), st)

    def fwd():
        lib.modegemm(xhat.data_ptr(), w.data_ptr(), gxhat.data_ptr(), st, **kw_f)

    def tf():                       # forward transform alone (k_fft2d_fwd3)
        lib.transform_forward(plan, _lib.SC_FWD_SCALED, x.data_ptr(), xhat.data_ptr(), B * C, ws.data_ptr(), st)

    def ti():                       # inverse transform alone (k_fft2d_inv3)
        lib.transform_inverse(plan, _lib.SC_INV_PADDED, xhat.data_ptr(), bias.data_ptr(), C, y.data_ptr(), B * C,
                              ws.data_ptr(), st)

    def gw_only():
        lib.modegemm(xhat.data_ptr(), ghat.data_ptr(), gw.data_ptr(), st, **kw_w)

    def gx_only():
        lib.modegemm(ghat.data_ptr(), w.data_ptr(), gxhat.data_ptr(), st, **kw_x)

    return dict(tf=tf, ti=ti, fwd=fwd, seq=seq, pair=pair, bwd=bwd, step=step, gw=gw_only, gx=gx_only)


fns = {name: make(name, lib) for name, lib in libs}
# correctness: pair == seq, bit for bit
for name, lib in libs:
    fns[name]["seq"]()
    torch.cuda.synchronize()
    r_w, r_x = gw.clone(), gxhat.clone()
    gw.fill_(float("nan")); gxhat.fill_(float("nan"))
    fns[name]["pair"]()
    torch.cuda.synchronize()
    print(f"{name:>10}: pair launch fused = {lib.modegemm_pair_fused(kw_w, kw_x)}; gW bits equal "
          f"{torch.equal(gw, r_w)}, gXhat bits equal {torch.equal(gxhat, r_x)}; kernels fwd/gW/gX "
          f"{lib.modegemm_path(**kw_f)}/{lib.modegemm_path(**kw_w)}/{lib.modegemm_path(**kw_x)}")
KINDS = tuple(os.environ.get("KINDS", "fwd,seq,pair,bwd,step").split(","))
res = {(n, k): [] for n, _ in libs for k in KINDS}
for _ in range(ROUNDS):
    for k in KINDS:
        for n, _ in libs:
            res[(n, k)].append(timed(fns[n][k]))
print(f"us per repetition (median / min of {ROUNDS} rounds x {REPS} reps), interleaved over the builds")
for n, _ in libs:
    row = []
    for k in KINDS:
        v = sorted(res[(n, k)])
        row.append(f"{k} {v[len(v) // 2]:7.1f} / {v[0]:7.1f}")
    print(f"{n:>10}: " + " | ".join(row))
