"""Autograd-aware collectives for the model-parallel region.

``scatter/gather/reduce/copy_*_model_parallel_region`` keep the contract of
/root/reference/neuralop/mpu/mappings.py:34-117 and helpers.py:102-166.  New here (the
reference defines ``_transpose``, helpers.py:81-99, but never calls it): the all-to-all
that re-shards a tensor between two of its dims -- the exchange step of the mode-parallel
spectral convolution.  One ``all_to_all_single`` per call (one contiguous message per peer:
xGMI is point-to-point, every peer pair has its own link, so an all-to-all is not ring-bound).
"""
import torch
import torch.distributed as dist

from .comm import get_model_parallel_group


def _size(group):
    return dist.get_world_size(group=group) if dist.is_initialized() else 1


# traffic counter (bench.py reports it): payload bytes this rank has handed to all_to_all_single
A2A_STATS = {"calls": 0, "bytes": 0}


def _a2a_issue(x, split_dim, group):
    """Start the exchange of ``x`` split into P chunks along split_dim (chunk p goes to rank p).  Returns
    (recv, work, send): recv is [P, *chunk shape] (real view of complex data), rank-major, valid once work is waited
    for.  ONE copy at most: the send buffer is the chunk-major permutation of x (a view when split_dim == 0)."""
    p = _size(group)
    if x.shape[split_dim] % p != 0:
        raise ValueError(f"dim {split_dim} of size {x.shape[split_dim]} not divisible by {p} ranks")
    xr = torch.view_as_real(x) if x.is_complex() else x
    send = xr.unflatten(split_dim, (p, xr.shape[split_dim] // p)).movedim(split_dim, 0).contiguous()
    recv = torch.empty_like(send)
    work = dist.all_to_all_single(recv, send, group=group, async_op=True)
    A2A_STATS["calls"] += 1
    A2A_STATS["bytes"] += send.numel() * send.element_size()
    return recv, work, send


def _a2a_finish(recv, cat_dim, is_complex):
    """[P, chunk...] -> the chunks concatenated along cat_dim in rank order (a view when cat_dim == 0)."""
    out = recv.movedim(0, cat_dim).flatten(cat_dim, cat_dim + 1)
    out = out.contiguous()
    return torch.view_as_complex(out) if is_complex else out


def _all_to_all(x, split_dim, cat_dim, group):
    """Split ``x`` into P chunks along split_dim, send chunk p to rank p, concatenate what
    arrives (in rank order) along cat_dim."""
    if _size(group) == 1:
        return x
    recv, work, _send = _a2a_issue(x, split_dim, group)
    work.wait()
    return _a2a_finish(recv, cat_dim, x.is_complex())


class _AllToAll(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, split_dim, cat_dim, group):
        ctx.split_dim, ctx.cat_dim, ctx.group = split_dim, cat_dim, group
        return _all_to_all(x, split_dim, cat_dim, group)

    @staticmethod
    def backward(ctx, g):
        # the adjoint of (split a, cat b) is (split b, cat a)
        return _all_to_all(g.contiguous(), ctx.cat_dim, ctx.split_dim, ctx.group), None, None, None


def all_to_all(x, split_dim, cat_dim, group=None):
    group = group if group is not None else get_model_parallel_group()
    return _AllToAll.apply(x, split_dim, cat_dim, group)


# ---- the reference's four region mappings ----------------------------------------------------
def _reduce(t, group):
    if _size(group) == 1:
        return t
    t = t.contiguous()
    dist.all_reduce(t, group=group)
    return t


def _split(t, dim, group):
    p = _size(group)
    if p == 1:
        return t
    if t.shape[dim] % p != 0:
        raise ValueError(f"cannot split dim {dim} of size {t.shape[dim]} evenly over {p} ranks")
    return t.chunk(p, dim=dim)[dist.get_rank(group=group)].contiguous()


def _gather(t, dim, group):
    p = _size(group)
    if p == 1:
        return t
    t = t.contiguous()
    parts = [torch.empty_like(t) for _ in range(p)]
    dist.all_gather(parts, t, group=group)
    return torch.cat(parts, dim=dim).contiguous()


class _Copy(torch.autograd.Function):
    @staticmethod
    def forward(ctx, t):
        return t

    @staticmethod
    def backward(ctx, g):
        return _reduce(g, get_model_parallel_group())


class _Reduce(torch.autograd.Function):
    @staticmethod
    def forward(ctx, t):
        return _reduce(t, get_model_parallel_group())

    @staticmethod
    def backward(ctx, g):
        return g


class _Scatter(torch.autograd.Function):
    @staticmethod
    def forward(ctx, t, dim):
        ctx.dim = dim
        return _split(t, dim, get_model_parallel_group())

    @staticmethod
    def backward(ctx, g):
        return _gather(g, ctx.dim, get_model_parallel_group()), None


class _Gather(torch.autograd.Function):
    @staticmethod
    def forward(ctx, t, dim):
        ctx.dim = dim
        return _gather(t, dim, get_model_parallel_group())

    @staticmethod
    def backward(ctx, g):
        return _split(g, ctx.dim, get_model_parallel_group()), None


def copy_to_model_parallel_region(t):
    return _Copy.apply(t)


def reduce_from_model_parallel_region(t):
    return _Reduce.apply(t)


def scatter_to_model_parallel_region(t, dim):
    return _Scatter.apply(t, dim)


def gather_from_model_parallel_region(t, dim):
    return _Gather.apply(t, dim)
