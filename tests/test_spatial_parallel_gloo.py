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
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


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
