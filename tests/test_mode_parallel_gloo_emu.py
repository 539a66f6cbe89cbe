"""world_size-2 gloo test (CPU) of the mode-parallel layer with the ENGINE ITSELF as the local stages: every transform,
contraction and gradient of each rank runs through the C-ABI of the host-emulation build of the kernels
(tests/emu_engine.py: the same sources, a thread per lane) -- the sharded-spectrum transforms writing / reading the
all-to-all buffer in place included -- where tests/test_mode_parallel_gloo.py injects the oracle's torch ops
(VERDICT r4 weak 1a).  Against the single-process oracle on the full batch.  Test infrastructure only: the product
refuses CPU tensors."""
import os
import socket
import sys

import pytest
import torch
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))


def _free_port():
    from neuraloperator_amd.mpu import comm
    return comm.free_port()                 # outside the ephemeral range: no client socket can be handed it


def _worker(rank, world, port, spatial, modes, bl, ci, co, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    torch.set_num_threads(1)
    from emu_engine import engine_on_emulation
    from neuraloperator_amd.modes import halve_last_mode
    from neuraloperator_amd.mpu import ModeParallelSpectralConv, comm
    from oracle import spectral_oracle as so

    comm.init(model_parallel_size=world, backend="gloo")
    nm = halve_last_mode(modes)
    B = bl * world
    torch.manual_seed(0)                      # identical full tensors on every rank
    x = torch.randn(B, ci, *spatial)
    g = torch.randn(B, co, *spatial)
    w = torch.empty(ci, co, *nm, dtype=torch.cfloat).normal_(0, 0.4)
    bias = torch.randn(co, *(1,) * len(spatial))
    with engine_on_emulation():
        conv = ModeParallelSpectralConv(ci, co, modes)            # default ops: the engine (here: its emulation build)
        from neuraloperator_amd.engine import EngineRawOps
        assert isinstance(conv.ops, EngineRawOps)
        with torch.no_grad():
            conv.weight.copy_(ModeParallelSpectralConv.shard_dense_weight(w, rank, world))
            conv.bias.copy_(bias)
        xs = x[rank * bl:(rank + 1) * bl].clone().requires_grad_(True)
        y = conv(xs)
        y.backward(g[rank * bl:(rank + 1) * bl])
        conv.reduce_replicated_grads()
    xf, wf, bf = x.clone().requires_grad_(True), w.clone().requires_grad_(True), bias.clone().requires_grad_(True)
    yf = so.forward_torch(xf, wf, bf, nm, nm)
    yf.backward(g)
    rows = -(-nm[0] // world)
    live = max(0, min(rows, nm[0] - rank * rows))
    ret[rank] = dict(
        y=so.rel_l2(y.detach().numpy(), yf.detach()[rank * bl:(rank + 1) * bl].numpy()),
        gx=so.rel_l2(xs.grad.numpy(), xf.grad[rank * bl:(rank + 1) * bl].numpy()),
        # (a rank whose block lies wholly past the kept rows holds inert zero rows: its weight gradient is zero)
        gw=so.rel_l2(conv.weight.grad[:, :, :live].numpy(), wf.grad[:, :, rank * rows:rank * rows + live].numpy())
        if live > 0 else float(conv.weight.grad.abs().max()),
        gb=so.rel_l2(conv.bias.grad.numpy(), bf.grad.numpy()),
    )
    comm.cleanup()


@pytest.mark.parametrize("spatial,modes,bl,ci,co", [
    ((16, 12), (8, 6), 2, 3, 4),        # size-agnostic passes, staging + permutation launch
    ((16, 12), (5, 6), 1, 3, 2),        # 5 mode rows over 2 ranks (a zero row on the wire), one sample per rank
    ((8, 8, 6), (4, 4, 4), 2, 2, 2),    # 3-d
    ((64, 256), (16, 12), 1, 2, 2),     # the fused 256-wide kernels address the sharded spectrum natively
])
def test_mode_parallel_on_the_emulated_engine(spatial, modes, bl, ci, co):
    _run(2, spatial, modes, bl, ci, co)


def test_mode_parallel_on_the_emulated_engine_world4():
    """four ranks, 6 mode rows over 4 blocks of 2 (the last block half empty: zero rows on the wire), one sample per rank"""
    _run(4, (16, 12), (6, 6), 1, 2, 3)


def test_mode_parallel_on_the_emulated_engine_world8():
    """BASELINE configs[3]'s layout in miniature: 8 ranks, one sample per rank, 4 of 32 first-dim rows per rank, 3-d"""
    _run(8, (32, 6, 8), (32, 4, 6), 1, 2, 2)


def _run(world, spatial, modes, bl, ci, co):
    from engine_runner import emu_lib
    emu_lib()                                   # build the emulation library once, before the workers race for it
    port = _free_port()
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    procs = [ctx.Process(target=_worker, args=(r, world, port, spatial, modes, bl, ci, co, ret)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(600)
        assert p.exitcode == 0, f"worker exit code {p.exitcode}"
    for r in range(world):
        assert all(v < 1e-5 for v in ret[r].values()), (r, dict(ret[r]))


def _variant_worker(rank, world, port, case, ret):
    """the rest of the plug-in contract on the sharded layer with the EMULATED ENGINE as local stages: runtime-reduced
    n_modes, a grid smaller than the modes, resolution changes, separable / Tucker / CP / TT weights, complex data"""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    torch.set_num_threads(1)
    from emu_engine import engine_on_emulation
    from neuraloperator_amd.modes import halve_last_mode
    from neuraloperator_amd.mpu import ModeParallelSpectralConv, comm
    from oracle import spectral_oracle as so

    spatial, max_modes, run_modes, out_shape, kind = case
    cplx, separable = kind == "complex", kind == "separable"
    fac = kind if kind in ("tucker", "cp", "tt") else None
    comm.init(model_parallel_size=world, backend="gloo")
    mx = halve_last_mode(max_modes, cplx)
    bl, ci = 1, 3
    B = bl * world
    co = ci if separable else 2
    torch.manual_seed(0)
    x = torch.randn(B, ci, *spatial, dtype=torch.cfloat if cplx else torch.float32)
    with engine_on_emulation():
        conv = ModeParallelSpectralConv(ci, co, max_modes, separable=separable, complex_data=cplx, factorization=fac,
                                        rank=0.6)
        torch.manual_seed(7)                                       # the same full (unsharded) parameters on every rank
        bias = torch.randn(co, *(1,) * len(spatial))
        full_shape = [ci, co, *mx]
        if fac is None:
            wf = torch.empty(*((ci,) if separable else (ci, co)), *mx, dtype=torch.cfloat).normal_(0, 0.4)
            sd = {"weight.tensor": wf, "bias": bias}
        elif fac == "tucker":
            ranks = list(conv.core.shape)
            core = torch.randn(*ranks, dtype=torch.cfloat) * 0.5
            facs = [torch.randn(n, r, dtype=torch.cfloat) * 0.7 for n, r in zip(full_shape, ranks)]
            sd = {"weight.core": core, "bias": bias, **{f"weight.factors.factor_{i}": f for i, f in enumerate(facs)}}
            wf = so.reconstruct_tucker(core, facs)
        elif fac == "cp":
            R = int(conv.cp_weights.shape[0])
            lam = torch.randn(R, dtype=torch.cfloat)
            facs = [torch.randn(n, R, dtype=torch.cfloat) * 0.6 for n in full_shape]
            sd = {"weight.weights": lam, "bias": bias, **{f"weight.factors.factor_{i}": f for i, f in enumerate(facs)}}
            wf = so.reconstruct_cp(lam, facs)
        else:
            ranks = [int(f.shape[0]) for f in conv.factors] + [1]
            cores = [torch.randn(ranks[i], n, ranks[i + 1], dtype=torch.cfloat) * 0.6 for i, n in enumerate(full_shape)]
            sd = {"bias": bias, **{f"weight.factors.factor_{i}": c for i, c in enumerate(cores)}}
            wf = so.reconstruct_tt(cores)
        conv.load_full_state_dict(sd)                              # an unsharded checkpoint: this rank keeps its rows
        if run_modes is not None:
            conv.n_modes = run_modes
        xs = x[rank * bl:(rank + 1) * bl].clone().requires_grad_(True)
        y = conv(xs, output_shape=out_shape)
        g = torch.randn(B, co, *y.shape[2:], generator=torch.Generator().manual_seed(5), dtype=y.dtype)
        y.backward(g[rank * bl:(rank + 1) * bl])
        conv.reduce_replicated_grads()
        n_modes_run = list(conv.n_modes)
    xf = x.clone().requires_grad_(True)
    yf = so.forward_torch(xf, wf, bias, n_modes_run, mx, separable=separable, output_shape=out_shape, complex_data=cplx)
    yf.backward(g)
    ret[rank] = dict(y=so.rel_l2(y.detach().numpy(), yf.detach()[rank * bl:(rank + 1) * bl].numpy()),
                     gx=so.rel_l2(xs.grad.numpy(), xf.grad[rank * bl:(rank + 1) * bl].numpy()))
    comm.cleanup()


@pytest.mark.parametrize("case", [
    ((16, 12), (8, 8), (6, 4), None, "dense"),          # n_modes lowered at run time
    ((6, 6), (8, 8), None, None, "dense"),              # grid smaller than the modes
    ((16, 12), (8, 6), None, (24, 20), "dense"),        # finer output grid
    ((16, 12), (8, 6), (6, 6), (12, 8), "dense"),       # fewer modes and a coarser output grid
    ((16, 12), (8, 6), None, None, "separable"),
    ((16, 12), (8, 6), None, None, "tucker"),
    ((16, 12), (8, 6), None, None, "cp"),
    ((16, 12), (8, 6), None, None, "tt"),
    ((12, 10), (8, 6), None, None, "complex"),
], ids=lambda c: f"{c[4]}_{'x'.join(map(str, c[0]))}_{c[2]}_{c[3]}")
def test_mode_parallel_variants_on_the_emulated_engine(case):
    from engine_runner import emu_lib
    emu_lib()
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    procs = [ctx.Process(target=_variant_worker, args=(r, world, port, case, ret)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(600)
        assert p.exitcode == 0, f"worker exit code {p.exitcode}"
    for r in range(world):
        assert all(v < 1e-5 for v in ret[r].values()), (r, dict(ret[r]))
