#!/usr/bin/env python
"""bench.py -- SpectralConv forward+backward throughput on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" = one forward + backward of ONE SpectralConv layer (grads for x, W, bias) over one
batch of synthetic fields already resident in HBM.  Default workload = BASELINE configs[1]:
B=32, C=64, 256x256, n_modes=(64,64) -> kept 64x33, fp32.  With N > 1 the default is the MODE-PARALLEL layer
(neuraloperator_amd/mpu: activations batch-sharded, B=32 per GPU -> weak scaling; weights sharded over the first
mode dim; RCCL all-to-all each way, pipelined in chunks) on the same workload, so that `value` measures the same
metric at every N; the line also carries `extra.dp_allreduce` (data-parallel replicas WITH the gradient
all-reduce of the 69 MB dense weight, what mode sharding avoids) and `extra.fno3d_modeshard` (BASELINE
configs[3]: 128^3, B=8 in total, strong scaling; its single-GPU number is `extra.fno3d_single` of the N=1 line).
The N=1 line also carries `extra.fno_block`: one whole FNO block forward + backward at the metric
shape, the reference's op sequence around the engine's convolution against the engine's fused passes; `extra.sfno`
(one SphericalConv); and `extra.fno3d_rank_of_8` (a child process: the per-rank step of an emulated 8-rank mode-parallel
group of configs[3] on this device -- timing only, the bound of its strong-scaling ratio this box can show).
--parallel replicas | modeshard | pencil selects one explicitly.

Rank 0 prints ONE JSON line (contract in the task statement) carrying `roofline` (dominant
kernel, timed live with events on the launch stream) and, at N=1, `cpu_baseline` (the
oracle's torch restatement of the reference CPU path on a bounded sample).

The timed region of the contract (W untimed steps, then exactly K steps between barrier + synchronize, max over
ranks) is run TWICE: straight after set-up (`cold_start` on the line) and again after --settle-ms of the same step
back to back (`clock_settle`); `value` / `ms_per_step` are the second, settled region.  MI355X clocks need ~20 ms of
sustained load to settle -- the first ~30 steps after any idle period run 10-25 % slower
(profiles/r02_clock_ramp.txt) -- so a 25-step region straight after start-up measures the governor, not the kernels
(see timed_steps and DESIGN.md 5).  --settle-ms 0 reports the cold region alone.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: 8 TB/s spec (6.29 TB/s measured copy)

WORKLOADS = {
    # name: (B, C, spatial, n_modes)
    "fno2d_256_m64_c64_b32": (32, 64, (256, 256), (64, 64)),      # BASELINE configs[1] (metric)
    "darcy_16_m12_c32_b4": (4, 32, (16, 16), (12, 12)),           # configs[0] shape
    "fno3d_128_m32_c32_b8": (8, 32, (128, 128, 128), (32, 32, 32)),  # configs[3] shape
    "fno2d_1024_m256_c128_b4": (4, 128, (1024, 1024), (256, 256)),   # configs[4] shape
    # not BASELINE configs: common small grids (plane kernels / two-pass route)
    "fno2d_64_m32_c64_b64": (64, 64, (64, 64), (32, 32)),
    "fno2d_128_m32_c64_b32": (32, 64, (128, 128), (32, 32)),
    "fno2d_192_m64_c64_b32": (32, 64, (192, 192), (64, 64)),      # radix-3 lines (32 x 6) on the two-pass route
    "fno3d_64_m16_c32_b8": (8, 32, (64, 64, 64), (16, 16, 16)),
    "fno2d_512_m64_c64_b8": (8, 64, (512, 512), (64, 64)),        # two-pass route, P = 16
    # the reference's documented Darcy-flow grids (doc/source/theory_guide/fno.rst:384-392: s = 85 / 141 / 211 / 421): lines
    # that are not 32 P points stay on the direct (pruned, matrix-core) DFT passes -- measured beside gpu_reference_baseline
    "darcy_85_m32_c32_b32": (32, 32, (85, 85), (32, 32)),
    "darcy_141_m32_c32_b32": (32, 32, (141, 141), (32, 32)),
    "darcy_211_m32_c32_b32": (32, 32, (211, 211), (32, 32)),
    "darcy_421_m32_c32_b16": (16, 32, (421, 421), (32, 32)),
    "darcy_421_m64_c32_b16": (16, 32, (421, 421), (64, 64)),
    # diagnostic: the per-rank transform load of configs[3] strong-scaled over 8 GPUs (one sample per rank)
    "fno3d_128_m32_c32_b1": (1, 32, (128, 128, 128), (32, 32, 32)),
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="fno2d_256_m64_c64_b32", choices=sorted(WORKLOADS))
    ap.add_argument("--parallel", default="auto", choices=["auto", "replicas", "modeshard", "pencil"],
                    help="N > 1: auto = modeshard; replicas = data-parallel copies with the gradient all-reduce; "
                         "modeshard = mode-parallel layer (batch-sharded activations, mode-sharded weights); pencil = "
                         "spatially decomposed layer (every sample spans all ranks: rows of the first grid dim sharded)")
    ap.add_argument("--comm-chunks", type=int, default=0,
                    help="modeshard: pieces every exchange is pipelined in (0 = the layer's default)")
    ap.add_argument("--chunk-dim", default=None, choices=["batch", "channels"],
                    help="modeshard: the chunked dim (default: batch, channels when a rank holds ONE sample).  "
                         "`--parallel modeshard --workload fno3d_128_m32_c32_b1 --chunk-dim channels` on ONE GPU = the "
                         "per-rank compute of configs[3] strong-scaled over 8 GPUs, chunked launches included "
                         "(a one-rank RCCL group: the exchanges degenerate to device copies)")
    ap.add_argument("--a2a", default="auto", choices=["auto", "torch", "native", "peer"],
                    help="modeshard: who moves the exchanges -- auto (torch.distributed, the engine's own RCCL binding inside "
                         "hipGraph steps), torch, native (mpu/rccl_native.py), peer (round 5, opt-in: direct stores into the "
                         "peers' HIP-IPC-mapped windows, mpu/peer_exchange.py; unmeasured on more than one GPU)")
    ap.add_argument("--emulate-world", type=int, default=1,
                    help="modeshard on ONE device, TIMING ONLY: the contraction of the mode-parallel layer runs on 1 / V of the "
                         "mode rows for V times the local batch -- the per-rank load of a V-rank group (transforms and the "
                         "degenerate exchanges are this rank's real ones; the numbers it computes are not the layer's)")
    ap.add_argument("--no-extras", action="store_true", help="skip the extra.* measurements")
    ap.add_argument("--no-gpu-reference", action="store_true", help="skip gpu_reference_baseline")
    ap.add_argument("--force-generic", action="store_true", help="A/B: skip the fused FFT kernels")
    ap.add_argument("--plan-flags", type=int, default=0, help="A/B: extra SC_PLAN_* bits OR-ed into the plan flags")
    ap.add_argument("--io", default="f32", choices=["f32", "bf16"],
                    help="storage type of the real tensors x / y / gy / gx (arithmetic is fp32 either way); "
                         "bf16 = the bf16-I/O reading of BASELINE configs[1], fused 2-D kernels only")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--graph", action="store_true",
                    help="time hipGraph replays of the step (neuraloperator_amd/graph.py) instead of eager steps: for "
                         "workloads whose eager step is bound by the host's issue rate.  Default (neither --graph nor "
                         "--no-graph): eager, except the mode-parallel layer with ONE sample per rank (configs[3] strong-scaled "
                         "over 8 GPUs: host-issue bound, DESIGN 6), which takes the native-RCCL hipGraph step when a probe "
                         "in a child process shows that this stack records and replays it (graph_probe)")
    ap.add_argument("--no-graph", action="store_true", help="never replay a hipGraph (A-B against the default)")
    ap.add_argument("--graph-probe", action="store_true",
                    help="internal: one rank of graph_probe (env RANK / WORLD_SIZE / LOCAL_RANK / MASTER_*): exit code 0 = "
                         "a mode-parallel step with native RCCL exchanges records into a hipGraph and replays bit-identically")
    ap.add_argument("--no-pmc", action="store_true",
                    help="roofline.traffic from profiles/pmc_traffic.json instead of two live rocprofv3 --pmc passes")
    ap.add_argument("--pmc-budget-s", type=float, default=300.0,
                    help="wall-clock budget shared by the live counter passes of the extra.* workloads")
    ap.add_argument("--settle-ms", type=float, default=SETTLE_MS,
                    help="untimed back-to-back steps (this many ms) between the cold timed region and the one `value` "
                         "reports: MI355X clocks settle ~20 ms into sustained load (0 = report the cold region only)")
    ap.add_argument("--stage-iters", type=int, default=10)
    ap.add_argument("--cpu-threads", type=int, default=0, help="internal: thread count of --cpu-baseline-only")
    ap.add_argument("--cpu-baseline-only", action="store_true",
                    help="internal: print the cpu_baseline JSON object and exit (no GPU touched)")
    return ap.parse_args()


def alg_bytes(B, C, spatial, kept, real_bytes=4):
    """R, Wb, S and BYTES_ALG = 4R + 3Wb + 9S per fwd+bwd step
    (real_bytes = 2 with bf16 I/O: the 4R term halves, 1592.8 MB at the metric shape)."""
    nsp = 1
    for s in spatial:
        nsp *= s
    mk = 1
    for k in kept:
        mk *= k
    R = real_bytes * B * C * nsp
    S = 8 * B * C * mk
    Wb = 8 * C * C * mk
    return R, Wb, S, 4 * R + 3 * Wb + 9 * S


def time_stage(fn, iters):
    fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True)
    e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters      # ms


def settle_clocks(fn, ms=100.0):
    """Repeat fn() back to back for ~ms of GPU time, untimed: every side measurement below is taken with settled
    clocks, like the headline (timed_steps explains why)."""
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    fn()
    e1.record()
    torch.cuda.synchronize()
    for _ in range(min(int(ms / max(e0.elapsed_time(e1), 1e-3)) + 1, 2000)):
        fn()


def device_copy_ceiling(nbytes, iters=10):
    """GB/s (read + written bytes) of a plain device-to-device copy of one real tensor: the practical
    ceiling a read-once / write-once pass can be held against on this box."""
    n = max(int(nbytes) // 4, 1 << 20)
    src = torch.empty(n, dtype=torch.float32, device="cuda").normal_()
    dst = torch.empty_like(src)
    settle_clocks(lambda: dst.copy_(src))
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        dst.copy_(src)
    e1.record()
    torch.cuda.synchronize()
    return round(2 * n * 4 * iters / (e0.elapsed_time(e1) * 1e-3) / 1e9, 1)


def stage_profile(B, C, spatial, n_modes, flags, iters, io="f32"):
    """Time every stage of the layer separately through the C-ABI (events on the launch
    stream) and return {stage: {ms, alg_bytes, GBs}} plus the plan's kernel names."""
    from neuraloperator_amd import _lib
    from neuraloperator_amd.engine import get_plan
    from neuraloperator_amd.modes import halve_last_mode, kept_block

    lib = _lib.get_lib()
    dev = torch.device("cuda", torch.cuda.current_device())
    nm = halve_last_mode(n_modes)
    kept, _ = kept_block(spatial, nm, nm)
    bf16 = io == "bf16"
    plan = get_plan(dev, spatial, kept, "forward", flags | (_lib.SC_PLAN_IO_BF16 if bf16 else 0))
    mk = 1
    for k in kept:
        mk *= k
    R, Wb, S, _ = alg_bytes(B, C, spatial, kept, 2 if bf16 else 4)
    x = torch.randn(B, C, *spatial, device=dev).to(torch.bfloat16 if bf16 else torch.float32)
    y = torch.empty_like(x)
    xh = torch.randn(B, C, mk, 2, device=dev)
    yh = torch.randn(B, C, mk, 2, device=dev)
    w = torch.randn(C, C, mk, 2, device=dev)
    gw = torch.empty_like(w)
    bias = torch.randn(C, device=dev)
    ws = torch.empty(lib.plan_workspace_bytes(plan, B * C) + 256, dtype=torch.uint8, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    p = lambda t: t.data_ptr()

    def gemm(a, b, c, **kw):
        lib.modegemm(p(a), p(b), p(c), st, n_modes=mk, **kw)

    xh2 = torch.empty_like(xh)
    kw_w = dict(P=C, Q=C, R=B, a_sp=mk, a_sr=C * mk, a_sm=1, conj_a=1, b_sr=C * mk, b_sq=mk, b_sm=1,
                c_sp=C * mk, c_sq=mk, c_sm=1, flags=_lib.SC_GEMM_STREAM_C)
    kw_x = dict(P=B, Q=C, R=C, a_sp=C * mk, a_sr=mk, a_sm=1, b_sr=mk, b_sq=C * mk, b_sm=1, conj_b=1,
                c_sp=C * mk, c_sq=mk, c_sm=1)
    paired = lib.modegemm_pair_fused(dict(kw_w, n_modes=mk), dict(kw_x, n_modes=mk))

    stages = {
        "fwd_transform": (lambda: lib.transform_forward(plan, _lib.SC_FWD_SCALED, p(x), p(xh), B * C, p(ws), st),
                          R + S),
        "contract_fwd": (lambda: gemm(xh, w, yh, P=B, Q=C, R=C, a_sp=C * mk, a_sr=mk, a_sm=1,
                                      b_sr=C * mk, b_sq=mk, b_sm=1, c_sp=C * mk, c_sq=mk, c_sm=1),
                         2 * S + Wb),
        "inv_transform": (lambda: lib.transform_inverse(plan, _lib.SC_INV_PADDED, p(yh), p(bias), C, p(y),
                                                        B * C, p(ws), st), R + S),
        "adj_c2r_transform": (lambda: lib.transform_forward(plan, _lib.SC_FWD_ADJ_C2R, p(x), p(xh), B * C,
                                                            p(ws), st), R + S),
        "contract_gw": (lambda: gemm(xh, yh, gw, **kw_w), 2 * S + Wb),
        "contract_gx": (lambda: gemm(yh, w, xh2, **kw_x), 2 * S + Wb),
        # what sc_layer_backward launches when the pair qualifies: both contractions (they share ghat) in ONE launch
        "contract_bwd": (lambda: lib.modegemm_pair(dict(kw_w, n_modes=mk), p(xh), p(yh), p(gw),
                                                   dict(kw_x, n_modes=mk), p(yh), p(w), p(xh2), st), 4 * S + 2 * Wb),
        "adj_r2c_transform": (lambda: lib.transform_inverse(plan, _lib.SC_INV_ADJ_R2C, p(yh), 0, C, p(y),
                                                            B * C, p(ws), st), R + S),
    }
    # the stages run in the layer's own order (fwd x3, bwd x4) with an event between each, so every
    # kernel sees the cache state it sees inside a real step (the 0.5 GB real tensors evict the
    # spectra / weights from L2 and Infinity Cache between uses); per-stage time = mean over iters
    order = ["fwd_transform", "contract_fwd", "inv_transform", "adj_c2r_transform"] + \
        (["contract_bwd"] if paired else ["contract_gw", "contract_gx"]) + ["adj_r2c_transform"]

    def sequence():
        for name in order:
            stages[name][0]()

    settle_clocks(sequence)
    # every iteration's events are recorded back to back and read after ONE synchronize at the end: a host-side
    # wait per iteration would idle the GPU between sequences (and let its clocks drop, timed_steps)
    evs = []
    for _ in range(iters):
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(len(order) + 1)]
        ev[0].record()
        for i, name in enumerate(order):
            stages[name][0]()
            ev[i + 1].record()
        evs.append(ev)
    torch.cuda.synchronize()
    acc = {name: 0.0 for name in order}
    for ev in evs:
        for i, name in enumerate(order):
            acc[name] += ev[i].elapsed_time(ev[i + 1])
    out = {}
    for name in order:
        ms = acc[name] / iters
        nbytes = stages[name][1]
        out[name] = {"ms": round(ms, 4), "alg_bytes": nbytes, "GBs": round(nbytes / ms / 1e6, 1)}
    # the four transform stages once more, each launched back to back between ONE pair of events: the in-sequence time
    # above includes the event records and the launch gap on both sides of the kernel (2-4 us), this one is the
    # kernel's own duration as rocprofv3 reports it (a transform streams a 0.5 GB real tensor with non-temporal
    # accesses, so repeating it changes nothing about its cache state; the contractions are not repeated this way:
    # back to back their 100 MB operands would stay in the Infinity Cache)
    for name in order:
        if not name.endswith("_transform"):
            continue
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        stages[name][0]()
        e0.record()
        for _ in range(2 * iters):
            stages[name][0]()
        e1.record()
        torch.cuda.synchronize()
        out[name]["ms_back_to_back"] = round(e0.elapsed_time(e1) / (2 * iters), 4)
    names = {"fwd": lib.plan_kernel_name(plan, 0), "inv": lib.plan_kernel_name(plan, 1),
             "fast": lib.plan_is_fast(plan)}
    return out, names


def block_extra(B, C, spatial, n_modes, dev):
    """one whole FNO block (fno_block.py:377-414, default configuration: linear skip, spectral
    convolution, GELU, ChannelMLP with expansion 0.5, soft-gating skip, GELU) forward + backward at the metric shape --
    the reference's op sequence with the engine's convolution inside (what a user of the conv_module plug-in runs)
    against the engine's three fused passes (neuraloperator_amd.blocks.fused_block_forward)."""
    import torch.nn.functional as F
    from torch import nn
    from neuraloperator_amd import SpectralConv, blocks as nb

    class Skip(nn.Module):                                   # skip_connections.py:119-169
        def __init__(self):
            super().__init__()
            self.conv = nn.Conv1d(C, C, 1, bias=False)

        def forward(self, x):
            return self.conv(x.reshape(x.shape[0], C, -1)).reshape(x.shape)
    Skip.__name__ = "Flattened1dConv"

    class Gate(nn.Module):                                   # skip_connections.py:53-117
        def __init__(self):
            super().__init__()
            self.weight, self.bias = nn.Parameter(torch.ones(1, C, *(1,) * len(spatial))), None

        def forward(self, x):
            return self.weight * x
    Gate.__name__ = "SoftGating"

    class MLP(nn.Module):                                    # channel_mlp.py:60-119
        def __init__(self):
            super().__init__()
            self.fcs = nn.ModuleList([nn.Conv1d(C, C // 2, 1), nn.Conv1d(C // 2, C, 1)])
            self.non_linearity, self.dropout = F.gelu, None

        def forward(self, x):
            return self.fcs[1](F.gelu(self.fcs[0](x.reshape(x.shape[0], C, -1)))).reshape(x.shape)

    class Block(nn.Module):
        def __init__(self):
            super().__init__()
            self.n_layers, self.non_linearity = 2, F.gelu
            self.preactivation, self.norm, self.stabilizer, self.complex_data, self.use_channel_mlp = False, None, None, False, True
            self.convs = nn.ModuleList([SpectralConv(C, C, n_modes)])
            self.fno_skips, self.channel_mlp, self.channel_mlp_skips = nn.ModuleList([Skip()]), nn.ModuleList([MLP()]), nn.ModuleList([Gate()])

        def forward(self, x, index=0, output_shape=None):
            y = F.gelu(self.convs[0](x) + self.fno_skips[0](x))
            return F.gelu(self.channel_mlp[0](y) + self.channel_mlp_skips[0](x))

    torch.manual_seed(7)
    blk = Block().to(dev)
    x = torch.randn(B, C, *spatial, device=dev, requires_grad=True)
    g = torch.randn(B, C, *spatial, device=dev)
    out = {"workload": f"one FNO block, B={B} C={C} {'x'.join(map(str, spatial))} modes {list(n_modes)}, forward + backward",
           "in_scope": bool(nb._block_in_scope(blk, 0, None))}
    for tag, fn in (("reference_op_sequence_ms", lambda t: blk(t, 0)), ("fused_ms", lambda t: nb.fused_block_forward(blk, t, 0))):
        def step():
            blk.zero_grad(set_to_none=True)
            x.grad = None
            fn(x).backward(g)
        settle_clocks(step)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            step()
        e1.record()
        torch.cuda.synchronize()
        out[tag] = round(e0.elapsed_time(e1) / 5, 4)
    return out


def sfno_extra(dev, B=8, C=32, nlat=128, nlon=256, L=64, M=64):
    """one SphericalConv (spherical_convolution.py:284-484: real SHT -> per-degree channel
    contraction -> inverse SHT) forward + backward on the engine's launches against the SAME operations as a torch op
    sequence on the GPU (rfft, two Legendre einsums with the layer's own tables, the channel einsum, irfft).  Parity
    against torch_harmonics itself is unpinned (absent); the restatement is pinned by closed-form harmonics on the CPU
    tier (tests/test_spherical.py)."""
    import math
    from neuraloperator_amd.spherical import SphericalConv, legendre_table, quadrature
    torch.manual_seed(5)
    conv = SphericalConv(C, C, (L, 2 * M), factorization=None, sht_grids="equiangular").to(dev)
    x = torch.randn(B, C, nlat, nlon, device=dev, requires_grad=True)
    g = torch.randn(B, C, nlat, nlon, device=dev)
    theta, wq = quadrature(nlat, "equiangular")
    cplx = lambda a: torch.from_numpy(a).to(torch.complex64).to(dev)
    Pf = cplx(legendre_table(M, L, theta, "ortho") * wq[None, None, :])                    # [m, l, k]
    Pi = cplx(legendre_table(M, L, theta, "ortho", inverse=True))

    def ref(t):
        X = torch.fft.rfft(t, dim=-1, norm="forward")[..., :M] * (2.0 * math.pi)           # (B, C, k, m)
        c = torch.einsum("bckm,mlk->bclm", X, Pf)
        y = torch.einsum("bilm,iol->bolm", c, conv.weight.tensor[..., :L])
        Y = torch.einsum("bolm,mlk->bokm", y, Pi)
        return torch.fft.irfft(Y, n=nlon, dim=-1, norm="forward") + conv.bias
    out = {"workload": f"SphericalConv B={B} C={C} {nlat}x{nlon} equiangular, l < {L}, m < {M}, dense weight, forward + backward"}
    with torch.no_grad():
        a, b = conv(x), ref(x)
        out["rel_l2_vs_op_sequence"] = float((a - b).norm() / b.norm())
    for tag, fn in (("op_sequence_ms", ref), ("engine_ms", conv)):
        def step():
            conv.zero_grad(set_to_none=True)
            x.grad = None
            fn(x).backward(g)
        settle_clocks(step)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            step()
        e1.record()
        torch.cuda.synchronize()
        out[tag] = round(e0.elapsed_time(e1) / 5, 4)
    return out


def _kernel_key(name):
    """'void k_fft2d_fwd3<256, float>(float const*, ...)' -> 'k_fft2d_fwd3<256, float>'"""
    name = name.replace("void ", "").strip()
    depth = 0
    for i, ch in enumerate(name):
        if ch == "<":
            depth += 1
        elif ch == ">":
            depth -= 1
        elif ch == "(" and depth == 0:
            return name[:i]
    return name


def measure_step_traffic(workload_shape, io="f32", kind="dense", reps=2, timeout_s=120):
    """HBM bytes of EVERY kernel of one layer step, read from the PMC counters IN THIS RUN: two rocprofv3 passes
    (FETCH_SIZE, WRITE_SIZE -- separate, as MI355X_MICROARCH.md prescribes; --kernel-trace only beside them) over
    scripts/layer_one.py (the whole step of that workload) in subprocesses; gfx950 correction FETCH_SIZE x 2,
    KB = 1024 B.  Returns (dict, note) or (None, reason):
        {"kernels": {name: {"launches_per_step", "fetch_B", "write_B", "traffic_B" (per launch), "ms" (mean duration
         under the counter pass: ranking only)}}, "step_traffic_B": sum over the step}
    rocprofv3 wraps a process, so the counters cannot be read inside the timed process itself."""
    import glob
    import shutil
    import sqlite3
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.isfile(exe):
        return None, "rocprofv3 not found"
    B, C, spatial, n_modes = workload_shape
    env = dict(os.environ, TMPDIR="/tmp", LAYER_REPS=str(reps), LAYER_IO=io, LAYER_KIND=kind,
               LAYER_SHAPE=",".join(str(v) for v in (B, C, *spatial, *n_modes)))
    vals, durs = {}, {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix="sc_pmc_", dir="/tmp")
        try:
            subprocess.run([exe, "--kernel-trace", "--pmc", counter, "-d", d, "-o", "run", "--", sys.executable,
                            os.path.join(ROOT, "scripts", "layer_one.py")], cwd="/tmp", env=env, timeout=timeout_s,
                           stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=True)
            got = {}
            for db in glob.glob(os.path.join(d, "**", "*.db"), recursive=True):
                cur = sqlite3.connect(db).cursor()
                for k, n, v in cur.execute("select kernel_name, counter_name, value from counters_collection"):
                    if n == counter and (k.startswith("k_") or k.startswith("void k_")):
                        got.setdefault(_kernel_key(k), []).append(float(v))
                if counter == "FETCH_SIZE":
                    try:
                        for k, calls, avg in cur.execute("select name, total_calls, average from top_kernels"):
                            durs[_kernel_key(k)] = float(avg) / 1e3          # top_kernels.average is in us
                    except sqlite3.Error:
                        pass
            if not got:
                return None, f"no {counter} rows"
            vals[counter] = got
        except Exception as e:                      # never let the counters kill the bench line
            return None, f"{counter} pass failed: {type(e).__name__}: {str(e)[:120]}"
        finally:
            shutil.rmtree(d, ignore_errors=True)
    kernels, step = {}, 0.0
    for k, f in vals["FETCH_SIZE"].items():
        w = vals["WRITE_SIZE"].get(k, [0.0])
        fb, wb = sum(f) / len(f) * 1024 * 2, sum(w) / len(w) * 1024
        kernels[k] = {"launches_per_step": round(len(f) / reps, 2), "fetch_B": int(fb), "write_B": int(wb),
                      "traffic_B": int(fb + wb), "ms": round(durs.get(k, 0.0), 4)}
        step += (fb + wb) * len(f) / reps
    return {"kernels": kernels, "step_traffic_B": int(step)}, \
        ("measured in this run: rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over "
         "scripts/layer_one.py, mean per launch, FETCH_SIZE x 2 (gfx950), KB = 1024 B")


def measure_traffic_live(workload_shape, kernel_substr, timeout_s=120):
    """the dominant kernel's HBM bytes per launch out of measure_step_traffic (+ the whole step's table)"""
    got, note = measure_step_traffic(workload_shape, timeout_s=timeout_s)
    if got is None:
        return None, note, None
    hits = [v for k, v in got["kernels"].items() if kernel_substr in k]
    if not hits:
        return None, f"no counter rows for {kernel_substr}", got
    return int(sum(h["traffic_B"] for h in hits) / len(hits)), note + f", mean of {kernel_substr} launches", got


PMC_BUDGET = {"left_s": 300.0}     # wall-clock budget of ALL live counter passes of one bench line (--pmc-budget-s)


def _extra_traffic(shape, io, kind, alg_bytes_step):
    """`traffic` block of an extra.* entry: the step's HBM bytes by the counters, their ratio to the algorithmic bytes and
    the kernel the step spends most of its time in (by launches x mean duration under the counter pass).  The passes of
    all extras share one wall-clock budget (each is two rocprofv3 subprocesses of up to 120 s): once it is
    spent the remaining entries say so instead of stretching the default line by minutes."""
    if PMC_BUDGET["left_s"] <= 0:
        return {"traffic": None, "traffic_note": "counter passes skipped: --pmc-budget-s spent"}
    t0 = time.perf_counter()
    got, note = measure_step_traffic(shape, io=io, kind=kind, timeout_s=max(min(120.0, PMC_BUDGET["left_s"]), 20.0))
    PMC_BUDGET["left_s"] -= time.perf_counter() - t0
    if got is None:
        return {"traffic": None, "traffic_note": note}
    ks = got["kernels"]
    dom = max(ks, key=lambda k: ks[k]["ms"] * ks[k]["launches_per_step"])
    return {"traffic": got["step_traffic_B"], "traffic_over_alg_bytes": round(got["step_traffic_B"] / alg_bytes_step, 3),
            "dominant_kernel": {"name": dom, **ks[dom]},
            "traffic_kernels": {k: v["traffic_B"] for k, v in sorted(ks.items(), key=lambda kv: -kv[1]["traffic_B"])[:8]},
            "traffic_source": note + "; `traffic` = sum over the step's launches"}


def rank_of_8_extra(steps, warmup, timeout_s=150):
    """configs[3]'s per-rank step of an 8-rank mode-parallel group on THIS device (a child process: `--parallel modeshard
    --emulate-world 8 --workload fno3d_128_m32_c32_b1`: one 128^3 sample per rank, the contraction load of one rank of eight, the exchanges degenerate) --
    TIMING ONLY: the kernel budget of a rank before real exchanges, i.e. the bound of configs[3]'s strong-scaling ratio
    that this box can show (`configs.c3_rank_of_8.bound_x` = one-GPU step / this).  A child process because the form needs a
    process group; whatever happens there never reaches the line's other numbers."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--parallel", "modeshard", "--emulate-world", "8", "--workload",
           "fno3d_128_m32_c32_b1", "--steps", str(steps), "--warmup", str(warmup), "--no-extras", "--no-cpu-baseline",
           "--no-gpu-reference", "--no-pmc"]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
        env.pop(k, None)
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout_s, env=env)
    except subprocess.TimeoutExpired:
        return {"ms_per_step": None, "note": f"did not finish within {timeout_s} s"}
    for line in reversed(r.stdout.strip().splitlines()):
        if line.startswith("{"):
            try:
                d = json.loads(line)
            except ValueError:
                break
            return {"ms_per_step": d.get("ms_per_step"), "cold_start_ms_per_step": d.get("cold_start", {}).get("ms_per_step"),
                    "launch": d.get("config", {}).get("launch"), "what": "one rank of an emulated 8-rank mode-parallel group, "
                    "B = 8 / 8 = one 128^3 sample per rank, timing only (bench.py --parallel modeshard --emulate-world 8)"}
    return {"ms_per_step": None, "note": "failed: " + (r.stderr.strip()[-200:] or "no JSON line")}


def compact_configs(out, extra, world):
    """The LAST key of the JSON line: every BASELINE config in <= 1500 characters, so that a record that
    keeps only the tail of the line still shows them.  Per entry: ms = ms per step of the settled region, cold = of the
    contract's first region, frac = algorithmic bytes / ms / 8 TB/s, toa = counter traffic / algorithmic bytes (null = not
    measured in this run), sps = samples/s of the whole job.  c1 = configs[1] (the headline workload, fp32 and bf16 real
    tensors), c2 = configs[2] TFNO Tucker rank 0.1, c3 = configs[3] FNO3d 128^3 B = 8 (one GPU: `c3_single`; N GPUs:
    `c3_modeshard` strong-scaled next to the one-GPU replica step and their ratio; one GPU: `c3_rank_of_8` = the per-rank
    step of an emulated 8-rank group, timing only, and the ratio it bounds), c4 = configs[4] 1024^2."""
    def ent(e, frac_key="frac_of_8TBs"):
        if not e or e.get("ms_per_step") is None:
            return None
        r = {"ms": e["ms_per_step"], "cold": e.get("cold_start_ms_per_step"), "sps": e.get("value")}
        if e.get(frac_key) is not None:
            r["frac"] = e[frac_key]
        if "traffic_over_alg_bytes" in e:
            r["toa"] = e["traffic_over_alg_bytes"]
        return r
    sr = out.get("step_roofline", {})
    head = {"ms": out["ms_per_step"], "cold": out["cold_start"]["ms_per_step"], "sps": out["value"],
            "frac": sr.get("frac_of_8TBs"), "toa": sr.get("traffic_over_alg_bytes")}
    io = out["config"].get("real_tensor_io", "f32")
    wl = out["config"]["workload"]
    c = {"n_gpus": world, "head": f"{wl}/{io}/{out['config'].get('parallelism')}"}
    key = {("fno2d_256_m64_c64_b32", "f32"): "c1_f32", ("fno2d_256_m64_c64_b32", "bf16"): "c1_bf16",
           ("fno3d_128_m32_c32_b8", "f32"): "c3_single", ("fno2d_1024_m256_c128_b4", "f32"): "c4_1024"}.get((wl, io), "head")
    par = str(out["config"].get("parallelism", ""))
    if world > 1 or par.startswith(("modeshard", "pencil")):   # N GPUs: the headline is a parallel form of its workload
        key = ("c3_" if wl.startswith("fno3d") else "c1_" if wl == "fno2d_256_m64_c64_b32" else "head_") + \
            (par.split("-")[0].rstrip("0123456789") or "parallel")
    c[key] = head
    for name, k in (("bf16_io", "c1_bf16"), ("tfno_rank01", "c2_tfno"), ("fno3d_single", "c3_single"),
                    ("fno2d_1024_b4", "c4_1024"), ("fno3d_modeshard", "c3_modeshard"), ("dp_allreduce", "c1_dp")):
        e = ent(extra.get(name))
        if e:
            c[k] = e
    if "c3_modeshard" in c and "c3_single" in c:           # configs[3]'s >= 6 x question on one line (strong scaling, B = 8 in all)
        c["c3_speedup"] = round(c["c3_single"]["ms"] / c["c3_modeshard"]["ms"], 3)
    r8 = extra.get("fno3d_rank_of_8")
    if r8 and r8.get("ms_per_step") and "c3_single" in c:  # one GPU: the per-rank kernel budget of 8 ranks and what it bounds
        c["c3_rank_of_8"] = {"ms": r8["ms_per_step"], "bound_x": round(c["c3_single"]["ms"] / r8["ms_per_step"], 2)}
    fb = extra.get("fno_block")
    if fb and fb.get("fused_ms") is not None:
        c["block"] = {"fused_ms": fb["fused_ms"], "ref_ms": fb.get("reference_op_sequence_ms")}
    sf = extra.get("sfno")
    if sf and sf.get("engine_ms") is not None:
        c["sfno"] = {"engine_ms": sf["engine_ms"], "ref_ms": sf.get("op_sequence_ms")}
    return c


def engine_path(names):
    """Which transform kernels the plan of this workload runs (sc_plan_kernel_name): the fused one-image-per-workgroup
    FFT (256-wide grids), the two-pass factorised FFT (512 / 1024 per axis), the factorised plane kernels (128 x 128 or 64 x 64
    last two axes) or the size-agnostic direct-DFT passes."""
    if names["fast"]:
        return "fused-fft"
    return {"k_f2p_r2c": "two-pass-fft", "k_pl128_fwd": "plane-fft", "k_pl64_fwd": "plane-fft"}.get(names["fwd"], "generic-dft")


def cpu_baseline(C, spatial, n_modes, threads, budget_s=10.0):
    """The reference's CPU path on the host cores of this box, fwd + autograd bwd of one SpectralConv layer, with
    ``threads`` torch threads.

    kind "reference": the VERBATIM module (oracle/ref_verbatim.py imports neuralop/layers/spectral_convolution.py from
    /root/reference through stubs of its absent third-party packages) -- possible only where the reference tree
    exists (the build container).  kind "port": oracle/spectral_oracle.forward_torch, the op-for-op torch
    restatement (rel-L2 0.0 against the verbatim module, tests/test_oracle_vs_reference.py) -- what runs on the GPU
    box, where /root/reference does not exist."""
    from oracle import spectral_oracle as so
    from neuraloperator_amd.modes import halve_last_mode

    cores = os.cpu_count() or 1
    torch.set_num_threads(threads)
    nm = halve_last_mode(n_modes)
    std = (2 / (2 * C)) ** 0.5
    torch.manual_seed(0)
    kind, ref_conv = "port", None
    try:
        from oracle import ref_verbatim
        if ref_verbatim.available():
            mod = ref_verbatim.load_reference()
            ref_conv = mod.SpectralConv(C, C, tuple(n_modes))
            kind = "reference"
    except Exception:
        ref_conv = None
    w = torch.empty(C, C, *nm, dtype=torch.cfloat).normal_(0, std).requires_grad_(True)
    bias = (std * torch.randn(C, *(1,) * len(spatial))).requires_grad_(True)
    b = 8                               # bounded sample of the B=32 workload (same C, grid, modes)
    x = torch.randn(b, C, *spatial, requires_grad=True)
    g = torch.randn(b, C, *spatial)

    def step():
        x.grad = None
        if ref_conv is not None:
            ref_conv.zero_grad(set_to_none=True)
            ref_conv(x).backward(g)
        else:
            w.grad = bias.grad = None
            so.forward_torch(x, w, bias, nm, nm).backward(g)

    step()                              # warm-up (thread pool, FFT plans, allocator)
    t0 = time.perf_counter()
    n = 0
    while n < 2 or (time.perf_counter() - t0 < budget_s and n < 5):
        step()
        n += 1
    dt = (time.perf_counter() - t0) / n
    what = "verbatim neuralop SpectralConv (oracle/ref_verbatim)" if kind == "reference" else "oracle.forward_torch"
    return {"value": round(b / dt, 3), "unit": "samples/s", "cores": threads, "kind": kind,
            "sample": f"B={b} of the same (C={C}, {'x'.join(map(str, spatial))}, modes {list(n_modes)}) workload, "
                      f"{n} timed fwd+bwd steps, torch {torch.__version__} CPU fp32, {what}, {threads} threads of "
                      f"{cores} cores"}


def cpu_baseline_subprocess(workload):
    """Run the CPU leg in fresh processes (no HIP runtime, own thread pool), each with a hard limit: 32 threads (where
    torch's intra-op pool stops scaling on this op chain) and, if it finishes in time, all cores.  The faster one is
    reported (`cores` = its thread count); the other is quoted in `sample`."""
    import subprocess
    cores = os.cpu_count() or 1
    runs = {}
    for threads, limit in ((min(cores, 32), 150), (cores, 45)):
        if threads in runs:
            continue
        try:
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-baseline-only", "--workload", workload,
                                "--cpu-threads", str(threads)], capture_output=True, text=True, timeout=limit)
            for line in reversed(r.stdout.strip().splitlines()):
                if line.startswith("{"):
                    runs[threads] = json.loads(line)
                    break
            else:
                runs[threads] = {"value": None, "sample": "failed: " + r.stderr.strip()[-160:]}
        except subprocess.TimeoutExpired:
            runs[threads] = {"value": None, "sample": f"{threads} threads: did not finish within {limit} s"}
    ok = {t: v for t, v in runs.items() if v.get("value")}
    if not ok:
        return {"value": None, "unit": "samples/s", "cores": 0, "kind": "port",
                "sample": "; ".join(str(v.get("sample")) for v in runs.values())}
    best = max(ok, key=lambda t: ok[t]["value"])
    out = dict(ok[best])
    others = [f"{t} threads: {v['value']} samples/s" if v.get("value") else str(v.get("sample"))
              for t, v in runs.items() if t != best]
    if others:
        out["sample"] += " (" + "; ".join(others) + ")"
    if best < cores:                                       # say it on the line
        out["sample"] += f" -- the host is NOT saturated ({best} of {cores} hardware threads): context, not a tuned CPU number"
    return out


def gpu_reference_baseline(B, C, spatial, n_modes, dev, steps=5):
    """The "before" number of SURVEY.md 8(d): the reference's own op chain (rfftn -> fftshift -> slice -> einsum ->
    zero-filled full spectrum -> ifftshift -> ifftn + irfft -> + bias, autograd backward) run through PyTorch-ROCm
    on this MI355X (hipFFT + ATen), same B / shapes / fp32.  oracle.forward_torch is that chain op for op (and the
    verbatim module itself where /root/reference exists)."""
    from oracle import spectral_oracle as so
    from neuraloperator_amd.modes import halve_last_mode

    nm = halve_last_mode(n_modes)
    std = (2 / (2 * C)) ** 0.5
    try:
        w = torch.empty(C, C, *nm, dtype=torch.cfloat, device=dev).normal_(0, std).requires_grad_(True)
        bias = (std * torch.randn(C, *(1,) * len(spatial), device=dev)).requires_grad_(True)
        x = torch.randn(B, C, *spatial, device=dev, requires_grad=True)
        g = torch.randn(B, C, *spatial, device=dev)

        def fwd(x_, w_, b_):
            # forward_torch allocates its zero spectrum on the CPU by default: run it under the device context
            with torch.device(dev):
                return so.forward_torch(x_, w_, b_, nm, nm)

        def step():
            x.grad = w.grad = bias.grad = None
            fwd(x, w, bias).backward(g)

        settle_clocks(step)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / steps * 1e3
        return {"value": round(B / (ms / 1e3), 2), "unit": "samples/s", "ms_per_step": round(ms, 3),
                "what": f"reference op chain (oracle.forward_torch = spectral_convolution.py:417-570 op for op) on "
                        f"PyTorch-ROCm {torch.__version__}: hipFFT + ATen einsum, autograd backward, B={B}, fp32"}
    except Exception as e:                      # never let the baseline leg kill the bench line
        return {"value": None, "unit": "samples/s", "what": f"failed: {type(e).__name__}: {str(e)[:160]}"}
    finally:
        torch.cuda.empty_cache()


SETTLE_MS = 250.0          # default of --settle-ms


def _timed_region(step, steps, warmup, dist, dev, share):
    """W untimed steps, then K steps between barrier + synchronize on both sides; max over ranks (ms per step).
    The interpreter's cyclic collector is held off for the region (a collection inside the 10-15 ms of a 20-step region
    shows up as 0.1-0.2 ms per step on the host-sensitive steps: the Tucker chain issues ~20 launches per step)."""
    import gc
    gc_was = gc.isenabled()
    gc.disable()                                           # (no collection here either: the device would idle behind it)
    try:
        for _ in range(warmup):
            step()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    finally:
        if gc_was:
            gc.enable()
    if dist is not None:
        t = torch.tensor([dt], device="cpu" if share else dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    return dt / steps * 1e3


def timed_steps(step, steps, warmup, dist, dev, share, settle_ms=SETTLE_MS):
    """The timed region of the contract (W untimed steps, then EXACTLY K steps between barrier + synchronize, max
    over ranks), run twice:

      cold    -- straight after set-up, as a process that has just started sees it;
      settled -- after `settle_ms` of the SAME step back to back (untimed), then again W untimed + K timed steps.

    Why: MI355X power management needs ~20 ms of sustained load before the clocks settle -- the first ~30 steps
    after ANY idle period (start-up, 0.2 s of host work) run 10-25 % slower than every later one
    (profiles/r02_clock_ramp.txt: 0.60-0.69 ms for the first blocks of 10 steps, 0.541-0.545 ms from step ~30 to
    step 600).  W = 5 and K = 20 steps of 0.55 ms all fall inside that ramp, so the cold number measures the
    governor, not the kernels, while a training run spends hours in the settled state.  Both numbers are on the
    JSON line (`cold_start`, `clock_settle`); `value` is the settled one.  The number of settling steps is derived
    from the cold time (already the max over ranks), hence identical on every rank of a collective step.
    Returns (ms_per_step settled, ms_per_step cold, settling steps)."""
    cold = _timed_region(step, steps, warmup, dist, dev, share)
    if settle_ms <= 0:
        return cold, cold, 0
    n = max(int(settle_ms / max(cold, 1e-3)) + 1, 1)
    for _ in range(n):
        step()
    return _timed_region(step, steps, warmup, dist, dev, share), cold, n


def build_case(parallel, workload, world, dev, flags, io_dtype, dist, seed, conv_kwargs=None, mp_kwargs=None):
    """(step fn, per-GPU batch, global batch, scaling, parallelism tag, conv) of one measurement.
    conv_kwargs: extra constructor arguments of the single-GPU layer (extra.tfno_rank01: Tucker weights)."""
    from neuraloperator_amd import SpectralConv

    B, C, spatial, n_modes = WORKLOADS[workload]
    torch.manual_seed(seed)
    local_spatial = list(spatial)
    post = None
    if parallel == "modeshard" and (world > 1 or dist is not None):
        from neuraloperator_amd.mpu import ModeParallelSpectralConv
        conv = ModeParallelSpectralConv(C, C, n_modes, engine_flags=flags, **(mp_kwargs or {})).to(dev)
        if workload == "fno2d_256_m64_c64_b32":
            b_local, scaling = B, "weak"                 # the metric workload: B = 32 per GPU at every N
        else:
            if B % world:
                return None
            b_local, scaling = B // world, "strong"      # configs[3]: the same 8 samples over N GPUs
        global_batch, par = b_local * world, f"modeshard{world}"
        post = conv.reduce_replicated_grads
    elif parallel == "pencil" and world > 1:
        from neuraloperator_amd.mpu import SpatialParallelSpectralConv
        if spatial[0] % world:
            raise SystemExit(f"--parallel pencil: grid rows {spatial[0]} not divisible by {world} ranks")
        conv = SpatialParallelSpectralConv(C, C, n_modes, engine_flags=flags).to(dev)
        b_local, scaling, global_batch, par = B, "strong", B, f"pencil{world}"
        local_spatial[0] = spatial[0] // world
    else:
        conv = SpectralConv(C, C, n_modes, engine_flags=flags, **(conv_kwargs or {})).to(dev)
        b_local, scaling, global_batch = B, "weak", B * world
        par = f"dp{world}-allreduce" if world > 1 else "single"
        if parallel == "replicas_single":                # one GPU's step, measured on every rank of an N-GPU run
            global_batch, par = B, "single (a replica per rank, no collective; value = ONE GPU's samples/s)"
        elif world > 1:
            def post():                                  # data parallel: the dense weight's gradient crosses xGMI
                for prm in conv.parameters():
                    if prm.grad is not None:
                        gr = torch.view_as_real(prm.grad) if prm.grad.is_complex() else prm.grad
                        dist.all_reduce(gr)
    x = torch.randn(b_local, C, *local_spatial, device=dev).to(io_dtype).requires_grad_(True)
    g = torch.randn(b_local, C, *local_spatial, device=dev).to(io_dtype)

    def step():
        x.grad = None
        for prm in conv.parameters():
            prm.grad = None
        y = conv(x)
        y.backward(g)
        if post is not None:
            post()

    step.x, step.g = x, g                                # --graph captures the same step on the same tensors
    return step, b_local, global_batch, scaling, par, conv


def free_port():
    from neuraloperator_amd.mpu import comm
    return comm.free_port()                              # outside the ephemeral range (see there)


def launch_command(n, argv, port):
    """The contract's own launch line for N ranks on one node (one rank per GPU, rendezvous on 127.0.0.1)."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
            "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


def self_launch(n, argv):
    """`python bench.py --gpus N` without a launcher around it (WORLD_SIZE unset): start the N ranks here, through
    the same `torch.distributed.run` line the driver uses, and hand its exit code on.  The children see WORLD_SIZE
    and take the ordinary path; rank 0's JSON line is the only thing on stdout (children point fd 1 at fd 2 until
    they print it; RCCL's banner and torchrun's notes go to stderr).  Fewer than N visible devices is an error, not
    a quiet one-rank run (SC_BENCH_SHARE_GPU=1: the test mode where every rank uses cuda:0 over gloo)."""
    import subprocess
    share = os.environ.get("SC_BENCH_SHARE_GPU") == "1"
    have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if have < (1 if share else n):
        print(f"bench.py: --gpus {n} needs {n} visible devices, found {have}", file=sys.stderr)
        return 2
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC only on this pool (RCCL needs it)
    env.setdefault("OMP_NUM_THREADS", str(max((os.cpu_count() or n) // n, 1)))
    cmd = launch_command(n, argv, free_port())
    print("[bench] starting", n, "ranks:", " ".join(cmd), file=sys.stderr, flush=True)
    return subprocess.call(cmd, env=env)


GRAPH_PROBE_TIMEOUT_S = 180


def graph_probe_child():
    """One rank of the probe (its own process: an abort, a segfault or a hang of this stack's capture path -- all three
    were seen in round 4, DESIGN 6 -- must cost the bench nothing but the graph).  The sequence is the bench's own:
    eager steps of the mode-parallel layer through torch.distributed under a NON-default stream, then the native RCCL
    communicator, then capture, then replays compared bit for bit with an eager step."""
    rank, local_rank = int(os.environ["RANK"]), int(os.environ["LOCAL_RANK"])
    world = int(os.environ["WORLD_SIZE"])
    import torch.distributed as dist
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    from neuraloperator_amd.graph import capture_step
    from neuraloperator_amd.mpu import ModeParallelSpectralConv, comm, rccl_native
    comm.init(model_parallel_size=world)
    side = torch.cuda.Stream()
    torch.cuda.set_stream(side)
    torch.manual_seed(7 + rank)
    rows = 2
    conv = ModeParallelSpectralConv(4, 4, (rows * world, 8, 8)).to(dev)
    conv.sync_replicated_parameters()
    x = torch.randn(1, 4, 2 * rows * world, 16, 16, device=dev, requires_grad=True)
    g = torch.randn(1, 4, 2 * rows * world, 16, 16, device=dev)
    params = [q for q in conv.parameters() if q.requires_grad]

    def eager():
        x.grad = None
        for q in params:
            q.grad = None
        y = conv(x)
        y.backward(g)
        conv.reduce_replicated_grads()
        torch.cuda.synchronize()
        return [y.detach().clone(), x.grad.clone()] + [torch.view_as_real(q.grad).clone() if q.grad.is_complex()
                                                       else q.grad.clone() for q in params]

    eager()                                                  # torch.distributed path, as the bench's earlier workloads
    rccl_native.prefer_native()
    if rccl_native.get(conv._group()) is None:
        print("[graph-probe] native RCCL path unavailable:", rccl_native.LAST_REASON, file=sys.stderr)
        return 3
    want = eager()                                           # native path, eager
    x.grad = None
    for q in params:
        q.grad = None
    st = capture_step(conv, x, g, post=conv.reduce_replicated_grads)
    for _ in range(3):
        st.replay()
    torch.cuda.synchronize()
    got = [st.output, x.grad] + [torch.view_as_real(q.grad) if q.grad.is_complex() else q.grad for q in params]
    ok = all(torch.equal(a, b) for a, b in zip(got, want))
    flag = torch.tensor([1 if ok else 0], device=dev, dtype=torch.int32)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    ok = int(flag.item()) == 1
    print(f"[graph-probe] rank {rank}: replay {'==' if ok else '!='} eager", file=sys.stderr)
    sys.stderr.flush()
    os._exit(0 if ok else 4)                                 # no teardown: ncclCommDestroy blocks once a graph recorded the communicator


def graph_probe(dist, world, rank, local_rank):
    """(ok, note), identical on every rank: every rank starts ONE child on its own GPU (a second process group on a port
    rank 0 draws), waits for it with a timeout, and the exit codes are MIN-reduced over the parents' group."""
    import subprocess
    box = [free_port() if rank == 0 else None]
    if world > 1:
        dist.broadcast_object_list(box, src=0)
    env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(local_rank),
               MASTER_ADDR="127.0.0.1", MASTER_PORT=str(box[0]))
    for k in [k for k in env if k.startswith("TORCHELASTIC_") or k in ("GROUP_RANK", "ROLE_RANK", "ROLE_WORLD_SIZE",
                                                                       "GROUP_WORLD_SIZE", "LOCAL_WORLD_SIZE")]:
        env.pop(k)            # (TORCHELASTIC_USE_AGENT_STORE would send the child's rendezvous to the parents' store)
    t0 = time.perf_counter()
    try:
        pr = subprocess.Popen([sys.executable, os.path.abspath(__file__), "--graph-probe"], env=env,
                              stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, start_new_session=True)
        try:
            _, err = pr.communicate(timeout=GRAPH_PROBE_TIMEOUT_S)
            lines = [ln for ln in err.decode(errors="replace").strip().splitlines() if "amdgpu.ids" not in ln]
            rc, tail = pr.returncode, lines[-1:]
            if rc != 0:                                  # the whole story goes to stderr, one line onto the bench line
                print(f"[bench] graph probe child of rank {rank} exited {rc}:\n  " + "\n  ".join(lines[-25:]), file=sys.stderr,
                      flush=True)
                first = [ln for ln in lines if "rror" in ln or "terminate" in ln or "what()" in ln]
                tail = first[:1] or tail
        except subprocess.TimeoutExpired:
            os.killpg(pr.pid, 9)                             # exactly the process group this call started
            pr.communicate()
            rc, tail = -9, [f"timeout after {GRAPH_PROBE_TIMEOUT_S} s"]
    except Exception as e:
        rc, tail = -1, [f"{type(e).__name__}: {e}"]
    flag = torch.tensor([1 if rc == 0 else 0], device="cuda", dtype=torch.int32)
    if world > 1:
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    ok = int(flag.item()) == 1
    note = f"probe {'ok' if ok else 'failed'} in {time.perf_counter() - t0:.1f} s" + \
        ("" if rc == 0 else f" (this rank: exit {rc}{': ' + tail[0][:120] if tail else ''})")
    return ok, note


def main():
    args = parse()
    if args.graph_probe:
        raise SystemExit(graph_probe_child())
    PMC_BUDGET["left_s"] = args.pmc_budget_s
    if args.cpu_baseline_only:
        B, C, spatial, n_modes = WORKLOADS[args.workload]
        print(json.dumps(cpu_baseline(C, spatial, n_modes, args.cpu_threads or min(os.cpu_count() or 1, 32))), flush=True)
        return
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(args.gpus, sys.argv[1:]))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world:
        # a line that says n_gpus = WORLD_SIZE while the caller asked for --gpus N would be a wrong scaling point
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    # stdout carries exactly ONE line, the JSON: everything libraries print on the way (RCCL's version banner on
    # communicator creation, gloo's connection notes) is sent to stderr by pointing fd 1 at fd 2 until that line
    sys.stdout.flush()
    stdout_fd = os.dup(1)
    os.dup2(2, 1)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU path)")
    # SC_BENCH_SHARE_GPU=1 (tests on a 1-GPU box only): every rank uses cuda:0 and all collectives run over gloo
    # (RCCL refuses two ranks on one device) -- exercises the launch contract (torchrun env, barriers,
    # max-over-ranks, rank-0 line) and the multi-rank data path without N devices.  Never set by the driver: the
    # real runs are one rank per GPU over RCCL.
    share = os.environ.get("SC_BENCH_SHARE_GPU") == "1"
    dev_index = 0 if share else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    dist = None
    if world > 1 or args.parallel == "modeshard":
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29512")
        if share:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        from neuraloperator_amd.mpu import comm
        comm.init(model_parallel_size=world)

    from neuraloperator_amd import _lib
    from neuraloperator_amd.modes import halve_last_mode, kept_block
    from neuraloperator_amd.mpu import mappings

    B, C, spatial, n_modes = WORKLOADS[args.workload]
    flags = (_lib.SC_PLAN_FORCE_GENERIC if args.force_generic else 0) | args.plan_flags
    io_dtype = torch.bfloat16 if args.io == "bf16" else torch.float32
    parallel = args.parallel
    if parallel == "auto":
        parallel = "modeshard" if world > 1 else "replicas"
    if share and parallel != "replicas" and args.io == "f32":
        pass                                             # gloo moves CUDA tensors through the host: slow but correct
    if args.io == "bf16":
        from neuraloperator_amd import engine
        kept_chk, _ = kept_block(spatial, halve_last_mode(n_modes), halve_last_mode(n_modes))
        if engine.get_plan_bf16_io(dev, list(spatial), kept_chk, "forward", flags) is None:
            raise SystemExit(f"--io bf16: {args.workload} does not run on the fused 2-D kernels (no bf16 I/O there)")

    if args.a2a != "auto":
        os.environ["SC_MPU_A2A"] = args.a2a               # read by mpu/rccl_native.py and mpu/peer_exchange.py
    if dist is not None and not share:
        # every step of a run with collectives is issued under a NON-default stream: on this stack a mode-parallel step
        # whose exchanges were ordered against torch's default (legacy null) stream makes a LATER capture of the layer
        # segfault in hipStreamEndCapture -- and the default line may capture after eager workloads ran
        torch.cuda.set_stream(torch.cuda.Stream(device=dev))
    if args.graph and dist is not None:
        from neuraloperator_amd.mpu import rccl_native
        rccl_native.prefer_native()                      # the exchanges must be plain stream launches to be recorded
    probe = {}

    def launch_form(case_x):
        """(step fn, launch tag) of a built case: eager, or hipGraph replays of the same step (see --graph)"""
        st_x, bl_x, _gb, _sc, tag_x, conv_x = case_x
        want = args.graph
        why = "--graph"
        if not want and not args.no_graph and dist is not None and not share and tag_x.startswith("modeshard") \
                and bl_x == 1:
            if "ok" not in probe:                        # once per run, the same answer on every rank
                probe["ok"], probe["note"] = graph_probe(dist, world, rank, local_rank)
                if rank == 0:
                    print("[bench] graph probe:", probe["note"], file=sys.stderr, flush=True)
            want, why = probe["ok"], "one sample per rank, " + probe["note"]
            if not want:
                return st_x, "eager (" + probe["note"] + ")"
        if not want:
            return st_x, "eager"
        from neuraloperator_amd.graph import capture_step
        post_g = None
        if dist is not None:
            # round 4: with the engine's native RCCL path (mpu/rccl_native.py) the exchanges of the mode-parallel layer
            # are plain stream-ordered launches and record into the graph; through torch.distributed they do not
            from neuraloperator_amd.mpu import rccl_native
            rccl_native.prefer_native()
            if not tag_x.startswith("modeshard") or rccl_native.get(conv_x._group()) is None:
                if args.graph:
                    raise SystemExit("--graph with collectives: the mode-parallel layer on the native RCCL path only "
                                     f"({rccl_native.LAST_REASON or tag_x})")
                return st_x, f"eager (native RCCL path unavailable: {rccl_native.LAST_REASON})"
            post_g = conv_x.reduce_replicated_grads
        try:
            gs = capture_step(conv_x, st_x.x, st_x.g, post=post_g)
        except Exception as e:                           # explicit --graph: an error; the default: the eager step
            if args.graph:
                raise
            print(f"[bench] rank {rank}: capture failed ({type(e).__name__}: {e}); eager step", file=sys.stderr)
            gs = None
        if dist is not None and world > 1:               # all ranks replay, or none does
            flag = torch.tensor([0 if gs is None else 1], device=dev, dtype=torch.int32)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            if int(flag.item()) != 1:
                gs = None
        if gs is None:
            return st_x, "eager (capture failed)"
        return gs.replay, f"hipGraph replay of the step ({why})"

    mp_kw = {}
    if args.comm_chunks > 0:
        mp_kw["comm_chunks"] = args.comm_chunks
    if args.chunk_dim:
        mp_kw["chunk_dim"] = args.chunk_dim
    if args.emulate_world > 1:
        if world != 1:
            raise SystemExit("--emulate-world: a one-device timing emulation (WORLD_SIZE must be 1)")
        mp_kw["emulate_world"] = args.emulate_world
    case = build_case(parallel, args.workload, world, dev, flags, io_dtype, dist, 1234 + rank, mp_kwargs=mp_kw)
    if case is None:
        raise SystemExit(f"{args.workload}: batch {B} not divisible by {world} ranks for the strong-scaling run")
    step, b_local, global_batch, scaling, par, conv = case
    mappings.A2A_STATS.update(calls=0, bytes=0)
    step, launch_tag = launch_form(case)
    # a replay does not pass through Python's exchange calls: what the capture (warm-up + recording) counted, per step
    a2a_rec = dict(mappings.A2A_STATS)
    graphed = launch_tag.startswith("hipGraph")
    mappings.A2A_STATS.update(calls=0, bytes=0)
    ms, ms_cold, n_settle = timed_steps(step, args.steps, args.warmup, dist, dev, share, args.settle_ms)
    value = global_batch / (ms / 1e3)
    a2a = dict(mappings.A2A_STATS)
    n_timed = (args.steps + args.warmup) * (2 if n_settle else 1) + n_settle
    del step, conv, case
    torch.cuda.empty_cache()

    # ---- extra measurements (same launch, same W + K protocol): see the module docstring
    extra = {}
    if not args.no_extras and args.workload == "fno2d_256_m64_c64_b32" and args.io == "f32":
        # every BASELINE config on the driver-timed line: name, parallelism, workload,
        # storage type of the real tensors, extra constructor arguments.  Each runs the contract's own W + K steps (after the same settling as the headline).
        f32, bf16 = torch.float32, torch.bfloat16
        tucker = dict(factorization="tucker", rank=0.1, implementation="factorized")
        todo = [("fno3d_single", "replicas", "fno3d_128_m32_c32_b8", f32, None),          # configs[3], one GPU
                ("tfno_rank01", "replicas", args.workload, f32, tucker),                  # configs[2]
                ("fno2d_1024_b4", "replicas", "fno2d_1024_m256_c128_b4", f32, None),      # configs[4]
                ("bf16_io", "replicas", args.workload, bf16, None),
                # the reference's full-resolution Darcy grid (421 points: no power-of-two / 32 P route applies; the
                # any-width matrix-core passes of round 4, profiles/r04_odd_sizes.txt)
                ("darcy_421", "replicas", "darcy_421_m32_c32_b16", f32, None)] if world == 1 else \
            [("dp_allreduce", "replicas", args.workload, f32, None),
             ("fno3d_modeshard", "modeshard", "fno3d_128_m32_c32_b8", f32, None),
             # the same B = 8 step on ONE GPU (every rank runs a replica, no collective): the denominator of the
             # strong-scaling ratio, measured in the same run on the same node (`configs.c3_speedup`)
             ("fno3d_single", "replicas_single", "fno3d_128_m32_c32_b8", f32, None)]
        for name, par_x, wl, io_x, kw_x in todo:
            if par_x == parallel and wl == args.workload and io_x == io_dtype and kw_x is None:
                continue
            try:
                c = build_case(par_x, wl, world, dev, flags, io_x, dist, 99 + rank, kw_x)
            except Exception as e:                       # an extra never kills the line
                extra[name] = {"value": None, "note": f"failed: {type(e).__name__}: {str(e)[:160]}"}
                continue
            if c is None:
                extra[name] = {"value": None, "note": f"batch of {wl} not divisible by {world} ranks"}
                continue
            st_x, bl_x, gb_x, sc_x, tag_x, conv_x = c
            st_x, launch_x = launch_form(c)
            ms_x, cold_x, n_x = timed_steps(st_x, args.steps, args.warmup, dist, dev, share, args.settle_ms)
            Bx, Cx, sp_x, nm_x = WORKLOADS[wl]
            kept_x, _ = kept_block(sp_x, halve_last_mode(nm_x), halve_last_mode(nm_x))
            Rx, Wbx, Sx, tot_x = alg_bytes(bl_x, Cx, sp_x, kept_x, 2 if io_x == bf16 else 4)
            formula = "4R+3Wb+9S"
            if kw_x is not None:                         # Tucker replaces 3 Wb by 3 (core + factors)
                wbytes = sum(8 * q.numel() for q in conv_x.weight.parameters())
                tot_x, formula = 4 * Rx + 3 * wbytes + 9 * Sx, "4R+3(core+factors)+9S"
                extra_cfg = {"weights": f"Tucker rank 0.1 -> core {list(conv_x.weight.core.shape)}, factorized contraction"}
            else:
                extra_cfg = {}
            gbs_x = tot_x / ms_x / 1e6                   # per GPU: bl_x samples' bytes per step time
            extra[name] = {"workload": wl, "parallelism": tag_x, "scaling": sc_x, "B_per_gpu": bl_x,
                           "global_batch": gb_x, "ms_per_step": round(ms_x, 4),
                           "value": round(gb_x / (ms_x / 1e3), 2), "unit": "samples/s", "steps": args.steps, "warmup": args.warmup,
                           "cold_start_ms_per_step": round(cold_x, 4), "settle_steps": n_x,
                           "real_tensor_io": "bf16" if io_x == bf16 else "f32", "launch": launch_x,
                           "alg_bytes_per_step": tot_x, "alg_bytes_formula": formula + " (SURVEY.md 8d), per GPU",
                           "achieved_GBs": round(gbs_x, 1), "frac_of_8TBs": round(gbs_x / HBM_PEAK_GBS, 4), **extra_cfg}
            if world > 1:                                # sharded weights: the single-GPU byte model does not apply
                for k in ("alg_bytes_per_step", "alg_bytes_formula", "achieved_GBs", "frac_of_8TBs"):
                    extra[name].pop(k)
            del st_x, conv_x, c
            torch.cuda.empty_cache()
            if world == 1 and not args.no_pmc:           # live counters for every BASELINE workload
                torch.cuda.synchronize()
                extra[name].update(_extra_traffic((bl_x, Cx, sp_x, nm_x), "bf16" if io_x == bf16 else "f32",
                                                  "tucker" if kw_x is not None else "dense", tot_x))
        if world == 1:
            extra["fno_block"] = block_extra(B, C, spatial, n_modes, dev)
            torch.cuda.empty_cache()
            try:
                extra["sfno"] = sfno_extra(dev)
            except Exception as e:                           # an extra never kills the line
                extra["sfno"] = {"engine_ms": None, "note": f"failed: {type(e).__name__}: {str(e)[:160]}"}
            torch.cuda.empty_cache()
            try:
                extra["fno3d_rank_of_8"] = rank_of_8_extra(args.steps, args.warmup)
            except Exception as e:                           # an extra never kills the line
                extra["fno3d_rank_of_8"] = {"ms_per_step": None, "note": f"failed: {type(e).__name__}: {str(e)[:160]}"}

    if rank == 0:
        nm = halve_last_mode(n_modes)
        kept, _ = kept_block(spatial, nm, nm)
        R, Wb, S, total = alg_bytes(b_local, C, spatial, kept, 2 if args.io == "bf16" else 4)
        stages, names = stage_profile(B if par.startswith(("modeshard", "pencil")) else b_local, C, spatial, n_modes,
                                      flags, args.stage_iters, args.io)
        dom = max(stages, key=lambda k: stages[k]["ms"])
        kern = names["fwd"] if dom in ("fwd_transform", "adj_c2r_transform") else \
            names["inv"] if dom in ("inv_transform", "adj_r2c_transform") else \
            "k_modegemm_dma_bwd" if dom == "contract_bwd" else "k_modegemm_dma"
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.isfile(tpath):
            try:
                tkey = args.workload + ("_bf16io" if args.io == "bf16" else "")
                traffic = json.load(open(tpath)).get(tkey, {}).get(dom) or None     # (0 = the regeneration did not see the kernel)
            except Exception:
                traffic = None
        # the dominant kernel's own duration: back-to-back launches where that is measured (transforms), else in sequence
        dom_ms = stages[dom].get("ms_back_to_back", stages[dom]["ms"])
        dom_gbs = round(stages[dom]["alg_bytes"] / dom_ms / 1e6, 1)
        traffic_source = ("profiles/pmc_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this kernel, gfx950 "
                          "corrections applied; not re-measured by this run)")
        step_tab = None
        if world == 1 and not args.no_pmc and args.io == "f32" and not args.plan_flags and not args.force_generic:
            torch.cuda.synchronize()
            live, note, step_tab = measure_traffic_live(WORKLOADS[args.workload], kern)
            if live is not None:
                traffic_lookup, traffic, traffic_source = traffic, live, note
                if traffic_lookup:
                    traffic_source += f"; committed lookup (profiles/pmc_traffic.json): {traffic_lookup}"
            else:
                traffic_source += f"; live counters unavailable ({note})"
        roof = {"bound": "hbm", "kernel": kern, "stage": dom,
                "achieved": dom_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(dom_gbs / HBM_PEAK_GBS, 4), "traffic": traffic,
                "traffic_source": traffic_source,
                "alg_bytes_per_launch": stages[dom]["alg_bytes"], "ms_per_launch": dom_ms,
                "ms_per_launch_how": "2 x stage-iters launches back to back between one pair of events"
                                     if "ms_back_to_back" in stages[dom] else "in the layer's sequence, an event each side"}
        step_gbs = total / ms / 1e6
        copy_gbs = device_copy_ceiling(R)
        roof["measured_copy_GBs"] = copy_gbs
        roof["frac_of_measured_copy"] = round(dom_gbs / copy_gbs, 4)
        out = {
            "metric": "FNO SpectralConv fwd+bwd samples/sec",
            "value": round(value, 2), "unit": "samples/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms, 4), "higher_is_better": True,
            "scaling": scaling, "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": args.workload, "B_per_gpu": b_local, "global_batch": global_batch,
                       "channels": C, "grid": list(spatial), "n_modes": list(n_modes), "kept": kept,
                       "parallelism": par, "engine_path": engine_path(names),
                       "launch": launch_tag,
                       **({"emulate_world": f"TIMING ONLY: contraction of a {args.emulate_world}-rank group's rank on one device"}
                          if args.emulate_world > 1 else {}),
                       "real_tensor_io": args.io,
                       "weights": "dense complex64, random init"},
            "roofline": roof,
            "step_roofline": {"alg_bytes_per_step": total, "achieved_GBs": round(step_gbs, 1),
                              "frac_of_8TBs": round(step_gbs / HBM_PEAK_GBS, 4),
                              # the same figure for the contract's FIRST timed region (before the clocks have settled)
                              "frac_of_8TBs_cold_start": round(total / ms_cold / 1e6 / HBM_PEAK_GBS, 4),
                              "frac_of_measured_copy": round(step_gbs / copy_gbs, 4),
                              "formula": "4R+3Wb+9S (SURVEY.md 8d), per GPU" +
                                         (", R at 2 bytes per value" if args.io == "bf16" else ""),
                              **({"traffic": step_tab["step_traffic_B"],
                                  "traffic_over_alg_bytes": round(step_tab["step_traffic_B"] / total, 3),
                                  "traffic_kernels": {k: v["traffic_B"] for k, v in step_tab["kernels"].items()}}
                                 if step_tab else {})},
            "stages": stages,
            "cold_start": {"ms_per_step": round(ms_cold, 4), "value": round(global_batch / (ms_cold / 1e3), 2),
                           "what": f"the same {args.warmup} untimed + {args.steps} timed steps straight after set-up, "
                                   "before the clocks have settled (first region; `value` is the second)"},
            "clock_settle": {"steps": n_settle, "ms": args.settle_ms,
                             "what": "untimed repetitions of the same step between the two timed regions "
                                     "(profiles/r02_clock_ramp.txt: the first ~30 steps after any idle period run "
                                     "10-25 % slower)"},
        }
        if world > 1 or parallel == "modeshard":
            from neuraloperator_amd.mpu import peer_exchange, rccl_native
            native = rccl_native.active()
            out["collectives"] = {"backend": "gloo (SC_BENCH_SHARE_GPU test mode)" if share else "nccl (RCCL over xGMI)",
                                  "issued_by": "peer stores into HIP-IPC-mapped windows, three engine launches per exchange "
                                               "(mpu/peer_exchange.py)" if peer_exchange.active() else
                                               "ncclAllToAll / ncclSend+ncclRecv straight on HIP streams (mpu/rccl_native.py)"
                                               if native else "torch.distributed" +
                                               (f" ({rccl_native.LAST_REASON})" if rccl_native.LAST_REASON else ""),
                                  # a replay runs the recorded exchanges: counted while the step was captured
                                  # (capture_step: 3 warm-up steps + the recording = 4 passes through the layer)
                                  "all_to_all_calls_per_step": round(a2a_rec["calls"] / 4, 2) if graphed
                                  else round(a2a["calls"] / n_timed, 2),
                                  "all_to_all_bytes_per_step_per_rank": int(a2a_rec["bytes"] / 4) if graphed
                                  else int(a2a["bytes"] / n_timed),
                                  "counted": "while the step was captured (recorded in the graph)" if graphed
                                  else "during the timed steps"}
        if extra:
            out["extra"] = extra
        if world == 1 and not args.no_gpu_reference and args.io == "f32":
            out["gpu_reference_baseline"] = gpu_reference_baseline(b_local, C, spatial, n_modes, dev)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline_subprocess(args.workload)
        out["configs"] = compact_configs(out, extra, world)   # LAST key: <= 1500 characters (tests/test_bench_contract.py)
        sys.stdout.flush()
        os.dup2(stdout_fd, 1)
        print(json.dumps(out), flush=True)
        os.dup2(2, 1)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
