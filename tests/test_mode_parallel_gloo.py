"""world_size-2 gloo test (CPU) of the mode-parallel layer: sharding, the two all-to-alls
and their autograd mirror, against the single-process oracle on the full batch.  The local
stages are the oracle's torch ops (tests/oracle_ops.py) -- the engine itself is GPU-only and
is covered by the -m gpu tier."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))


def _free_port():
    from neuraloperator_amd.mpu import comm
    return comm.free_port()                 # outside the ephemeral range: no client socket can be handed it


def _worker(rank, world, port, spatial, modes, ret, bl=2, chunks=4):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    torch.set_num_threads(2)
    from neuraloperator_amd.modes import halve_last_mode
    from neuraloperator_amd.mpu import ModeParallelSpectralConv, comm
    from oracle import spectral_oracle as so
    from oracle_ops import OracleRawOps

    comm.init(model_parallel_size=world, backend="gloo")
    assert comm.get_model_parallel_size() == world and comm.get_model_parallel_rank() == rank
    assert comm.get_data_parallel_size() == 1
    assert comm.get_global_rank() == rank and comm.get_world_rank() == rank
    from neuraloperator_amd.mpu.mappings import A2A_STATS
    nm = halve_last_mode(modes)
    B, ci, co = bl * world, 3, 4
    torch.manual_seed(0)                      # identical full tensors on every rank
    x = torch.randn(B, ci, *spatial)
    g = torch.randn(B, co, *spatial)
    w = torch.empty(ci, co, *nm, dtype=torch.cfloat).normal_(0, 0.4)
    bias = torch.randn(co, *(1,) * len(spatial))

    conv = ModeParallelSpectralConv(ci, co, modes, ops=OracleRawOps(nm), comm_chunks=chunks)
    if rank % 2:                                      # a reference / single-GPU checkpoint (real-view storage too)
        conv.load_full_state_dict({"conv.weight.tensor": torch.view_as_real(w), "conv.bias": bias}, prefix="conv.")
        assert torch.equal(conv.weight.detach(), ModeParallelSpectralConv.shard_dense_weight(w, rank, world))
    else:
        with torch.no_grad():
            conv.weight.copy_(ModeParallelSpectralConv.shard_dense_weight(w, rank, world))
            conv.bias.copy_(bias)
    xs = x[rank * bl:(rank + 1) * bl].clone().requires_grad_(True)
    A2A_STATS.update(calls=0, bytes=0)
    y = conv(xs)
    y.backward(g[rank * bl:(rank + 1) * bl])
    conv.reduce_replicated_grads()
    # 4 exchanges per layer step, each in `pieces` all-to-all calls, each moving the rank's whole
    # truncated spectrum once (zero-padded mode rows included): 4 S_local bytes per step and rank
    pieces = conv._chunks(bl >= 2)
    rows_ = -(-nm[0] // world)
    rest_ = int(np.prod(nm[1:]))
    n_fwd = min(pieces, bl if bl >= 2 else ci)
    n_bwd = min(pieces, bl if bl >= 2 else co)
    assert A2A_STATS["calls"] == 2 * (n_fwd + n_bwd), (A2A_STATS, n_fwd, n_bwd)
    assert A2A_STATS["bytes"] == 2 * 8 * bl * world * rows_ * rest_ * (ci + co), A2A_STATS

    xf, wf, bf = x.clone().requires_grad_(True), w.clone().requires_grad_(True), bias.clone().requires_grad_(True)
    yf = so.forward_torch(xf, wf, bf, nm, nm)
    yf.backward(g)
    rows = -(-nm[0] // world)                 # the last rank's shard may end in zero-padding rows
    live = min(rows, nm[0] - rank * rows)
    assert not conv.weight.grad[:, :, live:].abs().max() > 0 if live < rows else True
    errs = dict(
        y=so.rel_l2(y.detach().numpy(), yf.detach()[rank * bl:(rank + 1) * bl].numpy()),
        gx=so.rel_l2(xs.grad.numpy(), xf.grad[rank * bl:(rank + 1) * bl].numpy()),
        gw=so.rel_l2(conv.weight.grad[:, :, :live].numpy(), wf.grad[:, :, rank * rows:rank * rows + live].numpy()),
        gb=so.rel_l2(conv.bias.grad.numpy(), bf.grad.numpy()),
    )
    ret[rank] = errs
    comm.cleanup()


@pytest.mark.parametrize("spatial,modes,bl,chunks", [
    ((16, 12), (8, 6), 2, 4),          # even split, the local batch pipelined in two chunks
    ((8, 8, 6), (4, 4, 4), 2, 1),      # 3-d, one exchange per direction
    ((16, 12), (5, 6), 3, 2),          # 5 mode rows over 2 ranks: zero-padded rows on the wire, ragged batch chunks
    ((16, 12), (7, 6), 1, 2),          # one sample per rank: the exchange is pipelined over channel chunks
])
def test_mode_parallel_matches_single_process(spatial, modes, bl, chunks):
    _run_world(2, spatial, modes, bl, chunks)


@pytest.mark.parametrize("chunks", [None, 2, 3], ids=["default_one_piece", "two_channel_chunks", "three_ragged_chunks"])
def test_mode_parallel_world8_one_sample_per_rank(chunks):
    """BASELINE configs[3]'s layout on one node: 8 ranks, B = 8 in total (ONE sample per rank), 32 mode rows ->
    4 rows per rank, 3-d.  The default (None) exchanges each spectrum in one piece; 2 / 3 pipeline every exchange over
    channel chunks whose blocks are slabs of the contraction's operand / result (round 4: no copy around the
    collective, _Exchange.exchange_slabs; 3 = ragged chunks)."""
    _run_world(8, (32, 6, 8), (32, 4, 6), 1, chunks)


def _general_worker(rank, world, port, case, ret):
    """runtime-reduced n_modes / grid smaller than the modes / resolution change on the sharded layer"""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    torch.set_num_threads(2)
    from neuraloperator_amd.modes import halve_last_mode
    from neuraloperator_amd.mpu import ModeParallelSpectralConv, comm
    from oracle import spectral_oracle as so
    from oracle_ops import OracleAgOps, OracleAgOpsComplex, OracleRawOps

    spatial, max_modes, run_modes, out_shape, separable = case[:5]
    cplx = len(case) > 5 and bool(case[5])                         # complex_data=True (round 4)
    comm.init(model_parallel_size=world, backend="gloo")
    mx = halve_last_mode(max_modes, cplx)
    bl = 2
    B, ci = bl * world, 3
    co = ci if separable else 4
    torch.manual_seed(0)
    x = torch.randn(B, ci, *spatial, dtype=torch.cfloat if cplx else torch.float32)
    w = torch.empty(*((ci,) if separable else (ci, co)), *mx, dtype=torch.cfloat).normal_(0, 0.4)
    bias = torch.randn(co, *(1,) * len(spatial))
    conv = ModeParallelSpectralConv(ci, co, max_modes, ops=OracleRawOps(mx),
                                    agops=OracleAgOpsComplex() if cplx else OracleAgOps(), separable=separable,
                                    complex_data=cplx)
    conv.load_full_state_dict({"weight.tensor": w, "bias": bias})
    if run_modes is not None:
        conv.n_modes = run_modes                                   # fno_block.py:460-464
        assert conv.n_modes == halve_last_mode(run_modes, cplx) and conv.max_n_modes == mx
        with pytest.raises(ValueError):
            conv.n_modes = [m + 2 for m in max_modes]
        conv.n_modes = run_modes
    xs = x[rank * bl:(rank + 1) * bl].clone().requires_grad_(True)
    y = conv(xs, output_shape=out_shape)
    g = torch.randn(B, co, *y.shape[2:], generator=torch.Generator().manual_seed(5), dtype=y.dtype)
    y.backward(g[rank * bl:(rank + 1) * bl])
    conv.reduce_replicated_grads()

    xf, wf, bf = x.clone().requires_grad_(True), w.clone().requires_grad_(True), bias.clone().requires_grad_(True)
    yf = so.forward_torch(xf, wf, bf, conv.n_modes, mx, separable=separable, output_shape=out_shape, complex_data=cplx)
    yf.backward(g)
    rows = -(-mx[0] // world)
    live = min(rows, mx[0] - rank * rows)
    md = 1 if separable else 2
    gw_ref = wf.grad.narrow(md, rank * rows, live)
    gw = conv.weight.grad.narrow(md, 0, live)
    errs = dict(
        y=so.rel_l2(y.detach().numpy(), yf.detach()[rank * bl:(rank + 1) * bl].numpy()),
        gx=so.rel_l2(xs.grad.numpy(), xf.grad[rank * bl:(rank + 1) * bl].numpy()),
        gb=so.rel_l2(conv.bias.grad.numpy(), bf.grad.numpy()),
        gw=float((gw - gw_ref).abs().max() / max(float(gw_ref.abs().max()), 1e-30)),
    )
    # the skip path's resize is local: every rank resizes its own batch shard
    if out_shape is not None and len(spatial) == 2 and not cplx:
        t = conv.transform(xs.detach(), output_shape=out_shape)
        assert list(t.shape[2:]) == list(out_shape)
    ret[rank] = errs
    comm.cleanup()


@pytest.mark.parametrize("case", [
    ((16, 12), (8, 8), (6, 4), None, False),       # n_modes lowered at run time: rows 1..6 of the 8 stored rows
    ((16, 12), (8, 8), (5, 6), None, False),       # odd reduced count (python's floor on the negative slice bound)
    ((6, 6), (8, 8), None, None, False),           # grid smaller than the modes
    ((16, 12), (8, 6), None, (24, 20), False),     # finer output grid (the reference's end-padding placement)
    ((16, 12), (8, 6), (6, 6), (12, 8), False),    # fewer modes AND a coarser output grid
    ((8, 8, 6), (4, 4, 4), (2, 4, 4), None, False),
    ((16, 12), (8, 6), (6, 4), None, True),        # separable weights
    ((12, 10), (8, 6), None, None, False, True),   # complex_data=True (round 4): every dim centred, the last-dim quirk
    ((12, 10), (8, 6), (6, 4), None, False, True), # ... with n_modes lowered at run time
    ((8, 6, 10), (4, 4, 6), None, None, False, True),
], ids=["m8_to_6x4", "m8_to_5x6", "grid_smaller", "finer_out", "fewer_modes_coarser_out", "3d", "separable",
        "complex", "complex_fewer_modes", "complex_3d"])
def test_mode_parallel_general_path_matches_single_process(case):
    port = _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_general_worker, args=(2, port, case, ret), nprocs=2, join=True)
    assert len(ret) == 2
    for rank, errs in ret.items():
        for k, v in errs.items():
            assert np.isfinite(v) and v < 1e-5, (rank, k, v)


def _run_world(world, spatial, modes, bl, chunks):
    port = _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, port, spatial, modes, ret, bl, chunks), nprocs=world, join=True)
    assert len(ret) == world
    for rank, errs in ret.items():
        for k, v in errs.items():
            assert np.isfinite(v) and v < 1e-5, (rank, k, v)


def _a2a_worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    from neuraloperator_amd.mpu import (all_to_all, comm, gather_from_model_parallel_region,
                                        scatter_to_model_parallel_region)
    comm.init(model_parallel_size=world, backend="gloo")
    torch.manual_seed(1)
    full = torch.randn(4, 3, 6, 5, dtype=torch.cfloat)           # (B, C, k1, k2), identical on all ranks
    bl = 4 // world
    loc = full[rank * bl:(rank + 1) * bl].clone().requires_grad_(True)
    out = all_to_all(loc, split_dim=2, cat_dim=0)                # -> (B, C, k1/P, k2)
    rows = 6 // world
    ok_fwd = torch.equal(out.detach(), full[:, :, rank * rows:(rank + 1) * rows])
    gfull = torch.randn(4, 3, 6, 5, dtype=torch.cfloat)
    out.backward(gfull[:, :, rank * rows:(rank + 1) * rows])
    ok_bwd = torch.allclose(loc.grad, gfull[rank * bl:(rank + 1) * bl])
    # reference-style region mappings
    t = torch.arange(8.0).reshape(2, 4).requires_grad_(True)
    s = scatter_to_model_parallel_region(t, 1)
    gth = gather_from_model_parallel_region(s, 1)
    ok_sg = torch.equal(gth.detach(), t.detach())
    ret[rank] = (ok_fwd, ok_bwd, ok_sg)
    comm.cleanup()


def test_all_to_all_autograd_and_mappings():
    world = 2
    port = _free_port()
    ret = mp.Manager().dict()
    mp.spawn(_a2a_worker, args=(world, port, ret), nprocs=world, join=True)
    assert all(all(v) for v in ret.values()) and len(ret) == world


def test_single_process_fallback_getters():
    from neuraloperator_amd.mpu import comm
    assert comm.get_world_size() == 1 and comm.get_model_parallel_size() == 1
    assert comm.get_model_parallel_rank() == 0 and comm.get_data_parallel_rank() == 0


def _hybrid_worker(rank, world, port, ret):
    """4 ranks = 2 data-parallel replicas of a 2-rank mode-parallel group (contiguous model groups {0,1}, {2,3};
    strided data groups {0,2}, {1,3} -- reference comm.py:104-198): each replica sees half of the batch, the
    sharded weight gradient is summed over the DATA-parallel group only (never over the model group), the bias
    gradient over both."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    torch.set_num_threads(1)
    from neuraloperator_amd.modes import halve_last_mode
    from neuraloperator_amd.mpu import ModeParallelSpectralConv, comm
    from oracle import spectral_oracle as so
    from oracle_ops import OracleRawOps

    mp_size = 2
    comm.init(model_parallel_size=mp_size, backend="gloo")
    assert comm.get_model_parallel_size() == 2 and comm.get_data_parallel_size() == 2
    mp_rank, dp_rank = comm.get_model_parallel_rank(), comm.get_data_parallel_rank()
    assert (mp_rank, dp_rank) == (rank % 2, rank // 2)
    spatial, modes = (8, 10), (4, 6)
    nm = halve_last_mode(modes)
    B, ci, co = 8, 2, 3                      # global batch: 2 replicas x 2 model ranks x 2 samples
    torch.manual_seed(0)
    x = torch.randn(B, ci, *spatial)
    g = torch.randn(B, co, *spatial)
    w = torch.empty(ci, co, *nm, dtype=torch.cfloat).normal_(0, 0.4)
    bias = torch.randn(co, 1, 1)
    conv = ModeParallelSpectralConv(ci, co, modes, ops=OracleRawOps(nm), comm_chunks=2)
    with torch.no_grad():
        conv.weight.copy_(ModeParallelSpectralConv.shard_dense_weight(w, mp_rank, mp_size))
        conv.bias.copy_(bias)
    per = B // world
    lo = (dp_rank * mp_size + mp_rank) * per
    xs = x[lo:lo + per].clone().requires_grad_(True)
    conv(xs).backward(g[lo:lo + per])
    conv.reduce_replicated_grads()                                   # bias: over the model group
    dist_group = comm.get_data_parallel_group()
    dist.all_reduce(conv.weight.grad, group=dist_group)              # sharded weight: data-parallel group only
    dist.all_reduce(conv.bias.grad, group=dist_group)
    xf, wf, bf = x.clone().requires_grad_(True), w.clone().requires_grad_(True), bias.clone().requires_grad_(True)
    so.forward_torch(xf, wf, bf, nm, nm).backward(g)
    rows = nm[0] // mp_size
    ret[rank] = dict(
        gx=so.rel_l2(xs.grad.numpy(), xf.grad[lo:lo + per].numpy()),
        gw=so.rel_l2(conv.weight.grad.numpy(), wf.grad[:, :, mp_rank * rows:(mp_rank + 1) * rows].numpy()),
        gb=so.rel_l2(conv.bias.grad.numpy(), bf.grad.numpy()))
    comm.cleanup()


def test_hybrid_data_and_mode_parallel_groups():
    world = 4
    port = _free_port()
    ret = mp.Manager().dict()
    mp.spawn(_hybrid_worker, args=(world, port, ret), nprocs=world, join=True)
    assert len(ret) == world
    for rank, errs in ret.items():
        for k, v in errs.items():
            assert np.isfinite(v) and v < 1e-5, (rank, k, v)


def _tucker_worker(rank, world, port, spatial, modes, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    torch.set_num_threads(2)
    from neuraloperator_amd.modes import halve_last_mode
    from neuraloperator_amd.mpu import ModeParallelSpectralConv, comm
    from oracle import spectral_oracle as so
    from oracle_ops import OracleRawOps

    comm.init(model_parallel_size=world, backend="gloo")
    nm = halve_last_mode(modes)
    bl, ci, co = 2, 4, 3
    B = bl * world
    conv = ModeParallelSpectralConv(ci, co, modes, ops=OracleRawOps(nm), comm_chunks=2, factorization="tucker", rank=0.6)
    ranks = list(conv.core.shape)
    torch.manual_seed(0)                      # identical full tensors on every rank
    x = torch.randn(B, ci, *spatial)
    g = torch.randn(B, co, *spatial)
    core = torch.randn(*ranks, dtype=torch.cfloat) * 0.5
    full_f = [torch.randn(n, r, dtype=torch.cfloat) * 0.7 for n, r in zip([ci, co, *nm], ranks)]
    bias = torch.randn(co, *(1,) * len(spatial))
    with torch.no_grad():
        conv.core.copy_(core if rank == 0 else torch.zeros_like(core))       # rank 0 holds the values ...
        for i, f in enumerate(full_f):
            if i == 2:
                conv.factors[2].copy_(ModeParallelSpectralConv.shard_tucker_factor(f, rank, world))
            else:
                conv.factors[i].copy_(f if rank == 0 else torch.zeros_like(f))
        conv.bias.copy_(bias if rank == 0 else torch.zeros_like(bias))
    conv.sync_replicated_parameters(src=0)                                     # ... and broadcasts them
    # state-dict names follow tltorch's FactorList (a TFNO checkpoint's `weight.factors.factor_i`), and an unsharded
    # checkpoint loads through load_full_state_dict with this rank's rows of the first mode dim's factor
    assert {"core", "bias", *(f"factors.factor_{i}" for i in range(len(full_f)))} <= set(conv.state_dict())
    before = [f.detach().clone() for f in conv.factors]
    sd = {"weight.core": core, "bias": bias, **{f"weight.factors.factor_{i}": f for i, f in enumerate(full_f)}}
    conv.load_full_state_dict(sd)
    assert all(torch.equal(a, b.detach()) for a, b in zip(before, conv.factors)) and torch.equal(conv.core.detach(), core)
    xs = x[rank * bl:(rank + 1) * bl].clone().requires_grad_(True)
    y = conv(xs)
    y.backward(g[rank * bl:(rank + 1) * bl])
    conv.reduce_replicated_grads()

    xf = x.clone().requires_grad_(True)
    cf = core.clone().requires_grad_(True)
    ff = [f.clone().requires_grad_(True) for f in full_f]
    bf = bias.clone().requires_grad_(True)
    yf = so.forward_torch(xf, so.reconstruct_tucker(cf, ff), bf, nm, nm)
    yf.backward(g)
    rows = -(-nm[0] // world)
    live = min(rows, nm[0] - rank * rows)
    errs = dict(
        y=so.rel_l2(y.detach().numpy(), yf.detach()[rank * bl:(rank + 1) * bl].numpy()),
        gx=so.rel_l2(xs.grad.numpy(), xf.grad[rank * bl:(rank + 1) * bl].numpy()),
        gcore=so.rel_l2(conv.core.grad.numpy(), cf.grad.numpy()),
        gbias=so.rel_l2(conv.bias.grad.numpy(), bf.grad.numpy()),
        gshard=so.rel_l2(conv.factors[2].grad[:live].numpy(), ff[2].grad[rank * rows:rank * rows + live].numpy()),
    )
    for i in (0, 1, 3):
        errs[f"gfactor{i}"] = so.rel_l2(conv.factors[i].grad.numpy(), ff[i].grad.numpy())
    if live < rows:
        errs["pad_rows_zero"] = float(conv.factors[2].grad[live:].abs().max())
    ret[rank] = errs
    comm.cleanup()


@pytest.mark.parametrize("spatial,modes", [((16, 12), (8, 6)), ((16, 12), (5, 6))], ids=["even", "padded_rows"])
def test_mode_parallel_tucker_matches_single_process(spatial, modes):
    """TFNO weights in the mode-parallel layer: replicated core / channel / unsharded mode factors, the first mode
    dim's factor sharded by rows; output, input gradient and every parameter gradient (replicated ones after
    reduce_replicated_grads) equal the single-process oracle with the reconstructed dense weight."""
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_tucker_worker, args=(world, _free_port(), spatial, modes, ret), nprocs=world, join=True)
    assert len(ret) == world
    for rank, errs in ret.items():
        for k, v in errs.items():
            assert np.isfinite(v) and v < 2e-5, (rank, k, v)


def _variant_worker(rank, world, port, kind, spatial, modes, ret):
    """CP / TT / separable weights in the mode-parallel layer (round 3): the factor that carries the first mode dim (TT:
    the core; separable: the tensor itself) sharded by rows, everything else replicated."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    torch.set_num_threads(2)
    from neuraloperator_amd.modes import halve_last_mode
    from neuraloperator_amd.mpu import ModeParallelSpectralConv, comm
    from oracle import spectral_oracle as so
    from oracle_ops import OracleRawOps

    comm.init(model_parallel_size=world, backend="gloo")
    nm = halve_last_mode(modes)
    bl, ci, co = 2, 4, (4 if kind == "separable" else 3)
    B = bl * world
    kw = dict(separable=True) if kind == "separable" else dict(factorization=kind, rank=0.7)
    conv = ModeParallelSpectralConv(ci, co, modes, ops=OracleRawOps(nm), comm_chunks=2, **kw)
    rows = -(-nm[0] // world)
    live = min(rows, nm[0] - rank * rows)
    torch.manual_seed(0)                      # identical full tensors on every rank
    x = torch.randn(B, ci, *spatial)
    g = torch.randn(B, co, *spatial)
    bias = torch.randn(co, *(1,) * len(spatial))
    full_shape = [ci, co, *nm]
    if kind == "separable":
        wfull = torch.randn(ci, *nm, dtype=torch.cfloat) * 0.5
        sd = {"weight.tensor": wfull, "bias": bias}
        leaves = [wfull.clone().requires_grad_(True)]
        dense = lambda L: L[0]
    elif kind == "cp":
        R = int(conv.cp_weights.shape[0])
        lam = torch.randn(R, dtype=torch.cfloat)
        facs = [torch.randn(n, R, dtype=torch.cfloat) * 0.6 for n in full_shape]
        sd = {"weight.weights": lam, "bias": bias, **{f"weight.factors.factor_{i}": f for i, f in enumerate(facs)}}
        leaves = [lam.clone().requires_grad_(True)] + [f.clone().requires_grad_(True) for f in facs]
        dense = lambda L: so.reconstruct_cp(L[0], L[1:])
    else:
        ranks = [int(f.shape[0]) for f in conv.factors] + [1]
        cores = [torch.randn(ranks[i], n, ranks[i + 1], dtype=torch.cfloat) * 0.6 for i, n in enumerate(full_shape)]
        sd = {"bias": bias, **{f"weight.factors.factor_{i}": c for i, c in enumerate(cores)}}
        leaves = [c.clone().requires_grad_(True) for c in cores]
        dense = lambda L: so.reconstruct_tt(L)
    conv.load_full_state_dict(sd)             # an unsharded checkpoint: this rank keeps its rows
    xs = x[rank * bl:(rank + 1) * bl].clone().requires_grad_(True)
    y = conv(xs)
    y.backward(g[rank * bl:(rank + 1) * bl])
    conv.reduce_replicated_grads()

    xf, bf = x.clone().requires_grad_(True), bias.clone().requires_grad_(True)
    yf = so.forward_torch(xf, dense(leaves), bf, nm, nm, separable=(kind == "separable"),
                          contract=so.contract_dense_separable if kind == "separable" else so.contract_dense)
    yf.backward(g)
    errs = dict(y=so.rel_l2(y.detach().numpy(), yf.detach()[rank * bl:(rank + 1) * bl].numpy()),
                gx=so.rel_l2(xs.grad.numpy(), xf.grad[rank * bl:(rank + 1) * bl].numpy()),
                gb=so.rel_l2(conv.bias.grad.numpy(), bf.grad.numpy()))
    sl = slice(rank * rows, rank * rows + live)
    if kind == "separable":
        errs["gw"] = so.rel_l2(conv.weight.grad[:, :live].numpy(), leaves[0].grad[:, sl].numpy())
    else:
        params = ([conv.cp_weights] if kind == "cp" else []) + list(conv.factors)
        off = 1 if kind == "cp" else 0
        for i, (q, lf) in enumerate(zip(params, leaves)):
            if i - off == 2:                                   # the sharded factor / core: this rank's rows
                if kind == "tt":
                    errs[f"g{i}"] = so.rel_l2(q.grad[:, :live].numpy(), lf.grad[:, sl].numpy())
                else:
                    errs[f"g{i}"] = so.rel_l2(q.grad[:live].numpy(), lf.grad[sl].numpy())
            else:
                errs[f"g{i}"] = so.rel_l2(q.grad.numpy(), lf.grad.numpy())
    ret[rank] = errs
    comm.cleanup()


@pytest.mark.parametrize("kind,spatial,modes", [("cp", (16, 12), (8, 6)), ("tt", (16, 12), (8, 6)), ("separable", (16, 12), (8, 6)),
                                                ("cp", (16, 12), (5, 6)), ("tt", (8, 8, 6), (4, 4, 4)), ("separable", (16, 12), (7, 6))])
def test_mode_parallel_cp_tt_separable(kind, spatial, modes):
    world = 2
    ret = mp.Manager().dict()
    mp.spawn(_variant_worker, args=(world, _free_port(), kind, spatial, modes, ret), nprocs=world, join=True)
    assert len(ret) == world
    for rank, errs in ret.items():
        for k, v in errs.items():
            assert np.isfinite(v) and v < 2e-5, (kind, rank, k, v)


def _optim_worker(rank, world, port, ret):
    """Two AdamW steps on the mode-parallel layer: every rank's optimizer state is its own weight shard's (1/P of the
    dense weight's moments; the bias, replicated, steps identically on every rank after reduce_replicated_grads), and
    the stitched-together weights equal the single-process AdamW on the unsharded layer."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    torch.set_num_threads(2)
    from neuraloperator_amd import AdamW
    from neuraloperator_amd.modes import halve_last_mode
    from neuraloperator_amd.mpu import ModeParallelSpectralConv, comm
    from oracle import spectral_oracle as so
    from oracle_ops import OracleRawOps

    comm.init(model_parallel_size=world, backend="gloo")
    spatial, modes, bl, ci, co = (16, 12), (8, 6), 2, 3, 4
    nm = halve_last_mode(modes)
    rows = nm[0] // world
    torch.manual_seed(3)
    w = torch.empty(ci, co, *nm, dtype=torch.cfloat).normal_(0, 0.4)
    bias = torch.randn(co, 1, 1)
    xs = [torch.randn(bl * world, ci, *spatial) for _ in range(2)]
    gs = [torch.randn(bl * world, co, *spatial) for _ in range(2)]
    kw = dict(lr=1e-2, weight_decay=0.01)

    conv = ModeParallelSpectralConv(ci, co, modes, ops=OracleRawOps(nm))
    conv.load_full_state_dict({"weight": w, "bias": bias})
    opt = AdamW(conv.parameters(), **kw)
    wf, bf = torch.nn.Parameter(w.clone()), torch.nn.Parameter(bias.clone())
    optf = AdamW([wf, bf], **kw)
    for x, g in zip(xs, gs):
        opt.zero_grad()
        conv(x[rank * bl:(rank + 1) * bl]).backward(g[rank * bl:(rank + 1) * bl])
        conv.reduce_replicated_grads()
        opt.step()
        optf.zero_grad()
        so.forward_torch(x, wf, bf, nm, nm).backward(g)
        optf.step()
    st = opt.state[conv.weight]
    assert tuple(st["exp_avg"].shape) == (ci, co, rows, nm[1]) == tuple(conv.weight.shape)
    n_state = sum(v.numel() for s in opt.state.values() for v in s.values() if torch.is_tensor(v))
    n_full = sum(v.numel() for s in optf.state.values() for v in s.values() if torch.is_tensor(v))
    sl = slice(rank * rows, (rank + 1) * rows)
    ret[rank] = dict(
        w=so.rel_l2(conv.weight.detach().numpy(), wf.detach()[:, :, sl].numpy()),
        m=so.rel_l2(st["exp_avg"].numpy(), optf.state[wf]["exp_avg"][:, :, sl].numpy()),
        v=so.rel_l2(st["exp_avg_sq"].numpy(), optf.state[wf]["exp_avg_sq"][:, :, sl].numpy()),
        b=so.rel_l2(conv.bias.detach().numpy(), bf.detach().numpy()),
        state_numel=n_state, full_numel=n_full, bias_numel=bias.numel())
    comm.cleanup()


def test_optimizer_state_is_sharded_with_the_weight():
    """SURVEY 8 row f2 under mode parallelism: AdamW's moments live with the weight shard (no rank holds the dense
    weight's state), and two steps reproduce the single-process trajectory."""
    world = 2
    port = _free_port()
    ret = mp.Manager().dict()
    mp.spawn(_optim_worker, args=(world, port, ret), nprocs=world, join=True)
    assert len(ret) == world
    for rank, r in ret.items():
        for k in "wmvb":
            assert np.isfinite(r[k]) and r[k] < 1e-5, (rank, k, r[k])
        # per-rank state = the full state's weight part / world + the replicated bias's
        per_bias = 2 * r["bias_numel"]
        assert (r["state_numel"] - per_bias) * world == r["full_numel"] - per_bias, r
