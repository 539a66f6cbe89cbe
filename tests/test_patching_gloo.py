"""world_size-2 gloo test (CPU) of the distributed form of multigrid patching (patching.py:83-145): the stacked
patches are scattered over the model-parallel group along the batch dim, each rank runs the model on its share, the
outputs are gathered and stitched -- result and parameter gradient equal the single-process run."""
import contextlib
import io
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


def _worker(rank, world, port, stitching, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    torch.set_num_threads(2)
    import torch.distributed as dist
    from neuraloperator_amd.mpu import comm
    from neuraloperator_amd.mpu.patching import MultigridPatching2D

    comm.init(model_parallel_size=world, backend="gloo")
    torch.manual_seed(0)
    x = torch.randn(2, 2, 32, 32)
    y = torch.randn(2, 1, 32, 32)

    def run(distributed):
        torch.manual_seed(1)
        model = torch.nn.Conv2d(6, 1, 3, padding=1)               # 2 fine + 2 x 2 coarse channels at levels = 2
        with contextlib.redirect_stdout(io.StringIO()):
            pt = MultigridPatching2D(model, levels=2, padding_fraction=0.125, use_distributed=distributed,
                                     stitching=stitching)
        xp, yp = pt.patch(x, y)
        out, yt = pt.unpatch(model(xp), yp, evaluation=False)
        loss = ((out - yt) ** 2).sum()
        loss.backward()
        return out.detach(), yt.detach(), model.weight.grad.clone(), xp.shape[0]

    out_d, y_d, gw_d, nb_d = run(True)
    out_s, y_s, gw_s, nb_s = run(False)
    assert nb_d * world == nb_s                                  # patches scattered along the batch dim
    if stitching:
        # every rank holds the stitched field; the hook multiplies by the group size what each rank's share of the
        # patches contributed, the data-parallel mean (here: a sum / world) gives back the single-process gradient
        dist.all_reduce(gw_d)
        gw_d /= world
        ok = torch.allclose(out_d, out_s, atol=1e-6) and torch.allclose(gw_d, gw_s, rtol=1e-4, atol=1e-5)
    else:
        # un-stitched training: each rank keeps its own patches of output and target
        per = nb_s // world
        ok = torch.allclose(out_d, out_s[rank * per:(rank + 1) * per], atol=1e-6) and \
            torch.equal(y_d, y_s[rank * per:(rank + 1) * per])
        dist.all_reduce(gw_d)
        ok = ok and torch.allclose(gw_d, gw_s, rtol=1e-4, atol=1e-5)
    ret[rank] = bool(ok)
    comm.cleanup()


@pytest.mark.parametrize("stitching", [True, False])
def test_distributed_patching_matches_single_process(stitching):
    world = 2
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_worker, args=(world, _free_port(), stitching, ret), nprocs=world, join=True)
        assert all(ret.get(r) for r in range(world)), dict(ret)
