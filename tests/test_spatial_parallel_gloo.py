"""world_size-2 gloo test (CPU) of the spatially decomposed ("pencil") layer, SURVEY.md section 8 row f3: row
sharding, the column padding, the two all-to-alls and their autograd mirror, against the single-process oracle
on the full grid.  The local stages are the oracle's torch ops (tests/oracle_ops.py) -- the engine itself is
GPU-only; its (N-1)-d real plans and 1-d complex plans with frequency maps are covered by the -m gpu tier."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))


def _free_port():
    from neuraloperator_amd.mpu import comm
    return comm.free_port()                 # outside the ephemeral range: no client socket can be handed it


def _worker(rank, world, port, spatial, modes, batch, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    torch.set_num_threads(2)
    from neuraloperator_amd.modes import halve_last_mode
    from neuraloperator_amd.mpu import SpatialParallelSpectralConv, comm
    from oracle import spectral_oracle as so
    from oracle_ops import OracleOps

    comm.init(model_parallel_size=world, backend="gloo")
    nm = halve_last_mode(modes)
    B, ci, co = batch, 3, 4
    torch.manual_seed(0)                      # identical full tensors on every rank
    x = torch.randn(B, ci, *spatial)
    g = torch.randn(B, co, *spatial)
    w = torch.empty(ci, co, *nm, dtype=torch.cfloat).normal_(0, 0.4)
    bias = torch.randn(co, *(1,) * len(spatial))

    conv = SpatialParallelSpectralConv(ci, co, modes, ops=OracleOps(nm[1:]))
    with torch.no_grad():
        conv.weight.copy_(SpatialParallelSpectralConv.shard_dense_weight(w, rank, world))
        conv.bias.copy_(bias)
    hl = spatial[0] // world
    rows = slice(rank * hl, (rank + 1) * hl)
    xs = x[:, :, rows].clone().requires_grad_(True)
    y = conv(xs)
    y.backward(g[:, :, rows])
    conv.reduce_replicated_grads()

    xf, wf, bf = x.clone().requires_grad_(True), w.clone().requires_grad_(True), bias.clone().requires_grad_(True)
    yf = so.forward_torch(xf, wf, bf, nm, nm)
    yf.backward(g)
    gw_ref = SpatialParallelSpectralConv.shard_dense_weight(wf.grad, rank, world)   # padded columns: zero
    ret[rank] = dict(
        y=so.rel_l2(y.detach().numpy(), yf.detach()[:, :, rows].numpy()),
        gx=so.rel_l2(xs.grad.numpy(), xf.grad[:, :, rows].numpy()),
        gw=float(np.linalg.norm((conv.weight.grad - gw_ref).numpy().ravel()) /
                 np.linalg.norm(wf.grad.numpy().ravel())),
        gb=so.rel_l2(conv.bias.grad.numpy(), bf.grad.numpy()),
    )
    comm.cleanup()


@pytest.mark.parametrize("spatial,modes,batch", [((16, 12), (8, 6), 1),        # k2 = 4: even split
                                                 ((12, 10), (6, 8), 2),        # k2 = 5: padded to 6
                                                 ((8, 8, 6), (4, 4, 4), 1)])   # 3-d: local 2-d planes + axis pass
def test_spatial_parallel_matches_single_process(spatial, modes, batch):
    world = 2
    port = _free_port()
    ret = mp.Manager().dict()
    mp.spawn(_worker, args=(world, port, spatial, modes, batch, ret), nprocs=world, join=True)
    assert len(ret) == world
    for rank, errs in ret.items():
        for k, v in errs.items():
            assert np.isfinite(v) and v < 1e-5, (rank, k, v)


def test_single_rank_equals_dense_layer_maths():
    """P = 1 (no process group): the pipeline of (N-1)-d transform, axis pass, contraction and their inverses is
    the plain layer."""
    from neuraloperator_amd.modes import halve_last_mode
    from neuraloperator_amd.mpu import SpatialParallelSpectralConv
    from oracle import spectral_oracle as so
    from oracle_ops import OracleOps

    torch.manual_seed(3)
    spatial, modes = (10, 9), (5, 6)
    nm = halve_last_mode(modes)
    conv = SpatialParallelSpectralConv(2, 3, modes, ops=OracleOps(nm[1:]))
    x = torch.randn(2, 2, *spatial)
    y = conv(x)
    yo = so.forward_torch(x, conv.weight.detach(), conv.bias.detach(), nm, nm)
    assert so.rel_l2(y.detach().numpy(), yo.numpy()) < 1e-5


def _worker_variants(rank, world, port, spatial, modes, fac, out_shape, ret):
    """round 5: factorized weights (factors replicated, each rank reconstructs its mode columns) and a change of
    resolution (output_shape = the FULL output grid, whose first dim is sharded like the input's)"""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    torch.set_num_threads(2)
    from neuraloperator_amd.modes import halve_last_mode
    from neuraloperator_amd.mpu import SpatialParallelSpectralConv, comm
    from oracle import spectral_oracle as so
    from oracle_ops import PencilOracleOps

    comm.init(model_parallel_size=world, backend="gloo")
    nm = halve_last_mode(modes)
    B, ci, co = 2, 3, 4
    torch.manual_seed(100 + rank)             # a DIFFERENT init per rank: sync_replicated_parameters makes them one
    conv = SpatialParallelSpectralConv(ci, co, modes, ops=PencilOracleOps(), factorization=fac, rank=0.6)
    if fac != "dense":
        conv.sync_replicated_parameters()
        w = conv.weight.to_tensor().detach().clone()
    else:
        torch.manual_seed(5)
        w = torch.empty(ci, co, *nm, dtype=torch.cfloat).normal_(0, 0.4)
        with torch.no_grad():
            conv.weight.copy_(SpatialParallelSpectralConv.shard_dense_weight(w, rank, world))
        conv.sync_replicated_parameters()     # the bias
    bias = conv.bias.detach().clone()
    torch.manual_seed(0)                      # identical full tensors on every rank
    x = torch.randn(B, ci, *spatial)
    og = list(out_shape) if out_shape is not None else list(spatial)
    g = torch.randn(B, co, *og)
    hl, ho = spatial[0] // world, og[0] // world
    xs = x[:, :, rank * hl:(rank + 1) * hl].clone().requires_grad_(True)
    y = conv(xs, output_shape=out_shape)
    assert list(y.shape) == [B, co, ho, *og[1:]]
    y.backward(g[:, :, rank * ho:(rank + 1) * ho])
    conv.reduce_replicated_grads()

    xf, bf = x.clone().requires_grad_(True), bias.clone().requires_grad_(True)
    if fac != "dense":
        # single-process reference: the same factors (leaf copies), dense reconstruction, the oracle
        from neuraloperator_amd.factorized import SpectralWeight
        ref = SpectralWeight.new((ci, co, *nm), rank=0.6, factorization=fac)
        with torch.no_grad():
            for q, r in zip(ref.parameters(), conv.weight.parameters()):
                q.copy_(r)
        wf = ref.to_tensor()
    else:
        wf = w.clone().requires_grad_(True)
    yf = so.forward_torch(xf, wf, bf, nm, nm, output_shape=out_shape)
    yf.backward(g)
    errs = dict(
        y=so.rel_l2(y.detach().numpy(), yf.detach()[:, :, rank * ho:(rank + 1) * ho].numpy()),
        gx=so.rel_l2(xs.grad.numpy(), xf.grad[:, :, rank * hl:(rank + 1) * hl].numpy()),
        gb=so.rel_l2(conv.bias.grad.numpy(), bf.grad.numpy()),
    )
    if fac != "dense":
        for i, (q, r) in enumerate(zip(conv.weight.parameters(), ref.parameters())):
            errs[f"gfac{i}"] = so.rel_l2(torch.view_as_real(q.grad).numpy(), torch.view_as_real(r.grad).numpy())
    else:
        gw_ref = SpatialParallelSpectralConv.shard_dense_weight(wf.grad, rank, world)
        errs["gw"] = float(np.linalg.norm((conv.weight.grad - gw_ref).numpy().ravel()) /
                           np.linalg.norm(wf.grad.numpy().ravel()))
    ret[rank] = errs
    comm.cleanup()


@pytest.mark.parametrize("spatial,modes,fac,out_shape", [
    ((16, 12), (8, 6), "tucker", None),
    ((16, 12), (8, 6), "cp", None),
    ((16, 12), (6, 6), "tt", None),                 # k2 = 4 over 2 ranks
    ((8, 8, 6), (4, 3, 4), "tucker", None),         # 3-d, k2 = 3 over 2 ranks: a padded column
    ((16, 12), (8, 6), "dense", (24, 20)),          # finer output grid
    ((16, 12), (8, 6), "tucker", (12, 10)),         # coarser output grid, factorized weight
])
def test_spatial_parallel_variants(spatial, modes, fac, out_shape):
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    procs = [ctx.Process(target=_worker_variants, args=(r, world, port, spatial, modes, fac, out_shape, ret))
             for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0, f"worker exit code {p.exitcode}"
    for r in range(world):
        assert all(v < 1e-5 for v in ret[r].values()), (r, dict(ret[r]))


def test_padding_only_rank_builds_its_zero_block_on_the_weight_device_without_a_bias():
    """_local_weight() took the device of its zero block from the bias -- with bias=False the default (CPU)
    device against a GPU spectrum.  A padding-only rank (k2_pad / P * rank >= k2) of a factorized layer, checked on the
    `meta` device so that the CPU tier can tell the devices apart."""
    import torch.distributed as dist
    from neuraloperator_amd.mpu import SpatialParallelSpectralConv
    from oracle_ops import OracleOps
    port = _free_port()
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1)
    try:
        conv = SpatialParallelSpectralConv(3, 4, (8, 2), ops=OracleOps([2]), factorization="tucker", rank=0.6, bias=False,
                                           group=dist.group.WORLD)
        conv.P, conv.rank, conv.k2_pad, conv.k2_loc = 4, 3, 4, 1        # k2 = 2 over 4 ranks: rank 3 holds padding only
        conv = conv.to("meta")
        w = conv._local_weight()
        assert w.device.type == "meta" and list(w.shape) == [3, 4, 8, 1]
    finally:
        dist.destroy_process_group()


def _worker_general(rank, world, port, cfg, ret):
    """round 6: runtime n_modes, a grid smaller than the modes, complex_data, a change of resolution along any dim"""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    torch.set_num_threads(2)
    from neuraloperator_amd.modes import halve_last_mode
    from neuraloperator_amd.mpu import SpatialParallelSpectralConv, comm
    from oracle import spectral_oracle as so
    from oracle_ops import PencilOracleOps, PencilOracleOpsComplex

    comm.init(model_parallel_size=world, backend="gloo")
    spatial, modes, cplx, fac = cfg["spatial"], cfg["modes"], cfg.get("complex", False), cfg.get("fac", "dense")
    run_modes, out_shape, sep = cfg.get("run_modes"), cfg.get("out_shape"), cfg.get("separable", False)
    mx = halve_last_mode(modes, cplx)
    B, ci, co = 2, 3, (3 if sep else 4)
    lead = (ci,) if sep else (ci, co)
    dt = torch.cfloat if cplx else torch.float32
    torch.manual_seed(100 + rank)
    conv = SpatialParallelSpectralConv(ci, co, modes, ops=(PencilOracleOpsComplex if cplx else PencilOracleOps)(),
                                       factorization=fac, rank=0.6, complex_data=cplx, separable=sep)
    if fac != "dense":
        conv.sync_replicated_parameters()
    else:
        torch.manual_seed(5)
        w = torch.empty(*lead, *mx, dtype=torch.cfloat).normal_(0, 0.4)
        with torch.no_grad():
            conv.weight.copy_(SpatialParallelSpectralConv.shard_dense_weight(w, rank, world, separable=sep))
        conv.sync_replicated_parameters()     # the bias
    if run_modes is not None:
        conv.n_modes = run_modes
        assert conv.n_modes == halve_last_mode(run_modes, cplx) and conv.max_n_modes == mx
    nm = list(conv.n_modes)
    bias = conv.bias.detach().clone()
    torch.manual_seed(0)                      # identical full tensors on every rank
    x = torch.randn(B, ci, *spatial, dtype=dt)
    og = list(out_shape) if out_shape is not None else list(spatial)
    g = torch.randn(B, co, *og, dtype=dt)
    hl, ho = spatial[0] // world, og[0] // world
    xs = x[:, :, rank * hl:(rank + 1) * hl].clone().requires_grad_(True)
    y = conv(xs, output_shape=out_shape)
    assert list(y.shape) == [B, co, ho, *og[1:]] and y.is_complex() == cplx
    y.backward(g[:, :, rank * ho:(rank + 1) * ho])
    conv.reduce_replicated_grads()

    xf, bf = x.clone().requires_grad_(True), bias.clone().requires_grad_(True)
    if fac != "dense":
        from neuraloperator_amd.factorized import SpectralWeight
        ref = SpectralWeight.new((*lead, *mx), rank=0.6, factorization=fac)
        with torch.no_grad():
            for q, r in zip(ref.parameters(), conv.weight.parameters()):
                q.copy_(r)
        wf = ref.to_tensor()
    else:
        wf = w.clone().requires_grad_(True)
    yf = so.forward_torch(xf, wf, bf, nm, mx, output_shape=out_shape, complex_data=cplx, separable=sep)
    yf.backward(g)
    num = lambda t: torch.view_as_real(t.detach().contiguous()).numpy() if t.is_complex() else t.detach().numpy()
    errs = dict(
        y=so.rel_l2(num(y), num(yf[:, :, rank * ho:(rank + 1) * ho])),
        gx=so.rel_l2(num(xs.grad), num(xf.grad[:, :, rank * hl:(rank + 1) * hl])),
        gb=so.rel_l2(conv.bias.grad.numpy(), bf.grad.numpy()),
    )
    if fac != "dense":
        for i, (q, r) in enumerate(zip(conv.weight.parameters(), ref.parameters())):
            errs[f"gfac{i}"] = so.rel_l2(torch.view_as_real(q.grad).numpy(), torch.view_as_real(r.grad).numpy())
    else:
        gw_ref = SpatialParallelSpectralConv.shard_dense_weight(wf.grad, rank, world, separable=sep)
        errs["gw"] = float(np.linalg.norm((conv.weight.grad - gw_ref).numpy().ravel()) /
                           np.linalg.norm(wf.grad.numpy().ravel()))
    ret[rank] = errs
    comm.cleanup()


@pytest.mark.parametrize("cfg", [
    dict(spatial=(16, 12), modes=(8, 6), separable=True),                         # one (C, modes) weight
    dict(spatial=(8, 8, 6), modes=(6, 5, 4), separable=True, run_modes=(4, 3, 4), fac="cp"),
    dict(spatial=(16, 12), modes=(8, 6), run_modes=(6, 4)),                       # runtime n_modes, 2-d: fewer rows, fewer columns
    dict(spatial=(16, 12), modes=(8, 10), run_modes=(5, 7)),                      # odd row count, k2: 6 stored -> 4 used
    dict(spatial=(8, 8, 6), modes=(6, 5, 4), run_modes=(4, 3, 4)),                # 3-d: the sharded mode dim is a CENTRED one (offset 1)
    dict(spatial=(8, 8, 6), modes=(6, 6, 6), run_modes=(4, 2, 2), fac="tucker"),  # factorized weight, sub-block of the factors
    dict(spatial=(4, 6), modes=(8, 6)),                                           # the grid is smaller than the modes along dim 0
    dict(spatial=(16, 12), modes=(8, 6), complex=True),                           # complex data, 2-d
    dict(spatial=(8, 6, 6), modes=(4, 4, 3), complex=True),                       # complex data, 3-d
    dict(spatial=(16, 12), modes=(8, 6), complex=True, run_modes=(6, 3), out_shape=(24, 10)),
    dict(spatial=(8, 8, 6), modes=(4, 4, 4), out_shape=(8, 12, 6)),               # the MIDDLE dim changes (finer)
    dict(spatial=(8, 8, 6), modes=(6, 6, 4), out_shape=(12, 5, 10)),              # every dim changes, the middle one coarser than its modes
], ids=lambda c: "-".join(f"{k}={v}" for k, v in c.items()).replace(" ", ""))
def test_spatial_parallel_round6_variants(cfg):
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    procs = [ctx.Process(target=_worker_general, args=(r, world, port, cfg, ret)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0, f"worker exit code {p.exitcode}"
    for r in range(world):
        assert all(v < 1e-5 for v in ret[r].values()), (r, dict(ret[r]))
