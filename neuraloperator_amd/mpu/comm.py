"""Process-group wire-up for the mode-parallel layer.

Same surface as /root/reference/neuralop/mpu/comm.py:41-198 (``init``, ``get_world_size``,
``get_model_parallel_group/size/rank``, ``get_data_parallel_*``): contiguous model-parallel
groups of ``model_parallel_size`` ranks, strided data-parallel groups, and a size-1 / rank-0
fallback when torch.distributed is not initialised.  One process per GPU; backend "nccl"
(= RCCL over xGMI on ROCm) on GPUs, "gloo" for the CPU tests.
"""
import logging
import os

import torch
import torch.distributed as dist


class disable_logging:
    """Silence `logging` below `level` for the duration of a with-block (reference comm.py:24-32; the reference
    disables in the constructor and re-enables on exit, kept as is)."""

    def __init__(self, level=logging.ERROR):
        logging.disable(level=level)

    def __enter__(self):
        return self

    def __exit__(self, exc_type, exc, tb):
        logging.disable(level=logging.NOTSET)


_DATA_PARALLEL_GROUP = None
_MODEL_PARALLEL_GROUP = None
_MODEL_PARALLEL_SIZE = 1


def get_world_size():
    return dist.get_world_size() if dist.is_initialized() else 1


def get_world_rank():
    return dist.get_rank() if dist.is_initialized() else 0


def get_global_rank():
    """Reference comm.py:54-60: the world rank of this process, looked up through its data-parallel group (the
    reference passes its `get_local_rank()` -- which is dist.get_rank() there -- as the group rank; the only
    self-consistent reading is "my rank in the data-parallel group -> my world rank", which is what this returns)."""
    if not dist.is_initialized():
        return 0
    if _DATA_PARALLEL_GROUP is None:
        return dist.get_rank()
    return dist.get_global_rank(_DATA_PARALLEL_GROUP, dist.get_rank(group=_DATA_PARALLEL_GROUP))


def get_local_rank():
    if not dist.is_initialized():
        return 0
    return int(os.environ.get("LOCAL_RANK", get_world_rank() % max(torch.cuda.device_count(), 1)))


def get_data_parallel_size():
    return dist.get_world_size(group=_DATA_PARALLEL_GROUP) if dist.is_initialized() else 1


def get_data_parallel_rank():
    return dist.get_rank(group=_DATA_PARALLEL_GROUP) if dist.is_initialized() else 0


def get_data_parallel_group():
    return _DATA_PARALLEL_GROUP


def get_model_parallel_size():
    return dist.get_world_size(group=_MODEL_PARALLEL_GROUP) if dist.is_initialized() else 1


def get_model_parallel_rank():
    return dist.get_rank(group=_MODEL_PARALLEL_GROUP) if dist.is_initialized() else 0


def get_model_parallel_group():
    return _MODEL_PARALLEL_GROUP


def init(model_parallel_size=1, backend=None, verbose=False):
    """Initialise torch.distributed (if the launcher has not) and build the groups
    (reference comm.py:104-198).  Rendezvous comes from torchrun's env (RANK, WORLD_SIZE,
    LOCAL_RANK, MASTER_ADDR/PORT)."""
    global _DATA_PARALLEL_GROUP, _MODEL_PARALLEL_GROUP, _MODEL_PARALLEL_SIZE
    if not dist.is_initialized():
        if int(os.environ.get("WORLD_SIZE", "1")) == 1 and "RANK" not in os.environ:
            return                      # single process: getters fall back to size 1 / rank 0
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        kw = {}
        if backend == "nccl":
            local = int(os.environ.get("LOCAL_RANK", "0"))
            torch.cuda.set_device(local)
            kw["device_id"] = torch.device("cuda", local)
        dist.init_process_group(backend=backend, **kw)
    world = dist.get_world_size()
    rank = dist.get_rank()
    if world % model_parallel_size != 0:
        raise ValueError(f"world size {world} not divisible by model_parallel_size {model_parallel_size}")
    _MODEL_PARALLEL_SIZE = model_parallel_size
    n_model_groups = world // model_parallel_size
    model_groups = [list(range(i * model_parallel_size, (i + 1) * model_parallel_size))
                    for i in range(n_model_groups)]
    data_groups = [sorted(list(g)) for g in zip(*model_groups)]
    if verbose and rank == 0:
        print("model-parallel groups:", model_groups)
        print("data-parallel groups:", data_groups)
    for grp in data_groups:                       # every rank must create every group
        h = dist.new_group(ranks=grp)
        if rank in grp:
            _DATA_PARALLEL_GROUP = h
    for grp in model_groups:
        h = dist.new_group(ranks=grp)
        if rank in grp:
            _MODEL_PARALLEL_GROUP = h
    dist.barrier()


def cleanup():
    global _DATA_PARALLEL_GROUP, _MODEL_PARALLEL_GROUP
    _DATA_PARALLEL_GROUP = _MODEL_PARALLEL_GROUP = None
    from . import peer_exchange, rccl_native
    peer_exchange.shutdown()                  # unmap / free the peer windows (round 5) while the group still exists
    rccl_native.shutdown()                    # the engine's own RCCL communicators (round 4), before torch's
    if dist.is_initialized():
        dist.destroy_process_group()


def free_port():
    """A TCP port for a single-node rendezvous (MASTER_PORT), free right now and OUTSIDE the kernel's ephemeral range.

    A port taken from ``bind(("127.0.0.1", 0))`` lies INSIDE that range: between closing the probe socket and the
    store's own bind, any client socket of the same job (a rank connecting to the store, RCCL's bootstrap) can be
    handed the very same number as its source port -- the rendezvous then dies with EADDRINUSE (seen on the GPU tier,
    round 5: tests/test_gpu_graph.py on a busy box)."""
    import random
    import socket
    lo = 32768
    try:
        with open("/proc/sys/net/ipv4/ip_local_port_range") as f:
            lo = int(f.read().split()[0])
    except (OSError, ValueError, IndexError):
        pass
    hi = max(min(lo, 32768), 12000)
    rng = random.Random(os.getpid() * 7919 + int.from_bytes(os.urandom(4), "little"))
    for _ in range(200):
        port = rng.randrange(10000, hi)
        with socket.socket() as sk:
            try:
                sk.bind(("127.0.0.1", port))
            except OSError:
                continue
            return port
    raise RuntimeError("no free port below the ephemeral range")
