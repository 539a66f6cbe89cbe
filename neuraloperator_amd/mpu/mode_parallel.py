"""Mode-parallel SpectralConv: Fourier modes sharded over the model-parallel group.

New functionality on the reference's mpu API shape (the reference has
no mode-parallel conv, its ``_transpose`` all-to-all helper, mpu/helpers.py:81-99, is dead
code).  Layout per rank p of P:

    activations   batch-sharded   x_p   (B/P, Cin, d1..dN)            (like data parallel)
    weights       mode-sharded    W_p   (Cin, Cout, k1/P, k2..kN)     rows [p*k1/P, (p+1)*k1/P)

    x_p --pruned rFFT--> xhat_p (B/P, Cin, k1, ..) --all-to-all(split k1, cat batch)-->
    (B, Cin, k1/P, ..) --contract with W_p--> (B, Cout, k1/P, ..)
    --all-to-all(split batch, cat k1)--> (B/P, Cout, k1, ..) --zero-padded inverse--> y_p

Each rank sees the whole batch for its modes, so gW needs NO all-reduce (a dense layer's
69 MB weight gradient would be ring-bound on xGMI); the bias is replicated and its gradient
is summed over the group.  The backward is the same pipeline mirrored (2 more all-to-alls).
"""
import torch
from torch import nn

from ..modes import halve_last_mode, kept_block
from ..spectral_conv import BaseSpectralConv
from . import comm
from .mappings import all_to_all


class ModeParallelSpectralConv(BaseSpectralConv):
    """Dense-weight SpectralConv whose first mode dim is sharded across the model-parallel group.

    Constructor arguments follow SpectralConv; ``n_modes`` is fixed at construction (the
    shard layout depends on it).  ``ops`` (tests only) replaces the three local stages."""

    def __init__(self, in_channels, out_channels, n_modes, bias=True, init_std="auto",
                 fft_norm="forward", device=None, engine_flags=0, group=None, ops=None, **unused):
        super().__init__(device=device)
        for k in ("complex_data", "separable"):
            if unused.get(k):
                raise NotImplementedError(f"{k}=True is not supported by the mode-parallel layer")
        if unused.get("factorization") not in (None, "Dense", "dense"):
            raise NotImplementedError("mode-parallel layer: dense weights only")
        self.in_channels, self.out_channels = in_channels, out_channels
        self._n_modes = halve_last_mode(n_modes)
        self.max_n_modes = list(self._n_modes)
        self.order = len(self._n_modes)
        if self.order < 2:
            raise NotImplementedError("mode sharding needs >= 2 spatial dims (dim 0 is sharded)")
        self.fft_norm = fft_norm
        self.group = group
        self.P = comm.get_model_parallel_size() if group is None else torch.distributed.get_world_size(group)
        self.rank = comm.get_model_parallel_rank() if group is None else torch.distributed.get_rank(group)
        if self._n_modes[0] % self.P != 0:
            raise ValueError(f"n_modes[0]={self._n_modes[0]} must be divisible by the {self.P} model-parallel ranks")
        if init_std == "auto":
            init_std = (2 / (in_channels + out_channels)) ** 0.5
        rows = self._n_modes[0] // self.P
        w = torch.empty(in_channels, out_channels, rows, *self._n_modes[1:], dtype=torch.cfloat, device=device)
        w.normal_(0, init_std)
        self.weight = nn.Parameter(w)
        self.weight.mode_sharded = True          # exclude from data-parallel all-reduce within the group
        self.bias = nn.Parameter(init_std * torch.randn(out_channels, *(1,) * self.order, device=device)) \
            if bias else None
        if ops is None:
            from ..engine import EngineOps
            ops = EngineOps(fft_norm, engine_flags)
        self.ops = ops

    @property
    def n_modes(self):
        return self._n_modes

    @n_modes.setter
    def n_modes(self, value):
        raise NotImplementedError("the mode-parallel layer fixes n_modes at construction (shard layout)")

    def transform(self, x, output_shape=None):
        if output_shape is not None and list(output_shape) != list(x.shape[2:]):
            raise NotImplementedError("resolution change is not supported by the mode-parallel layer")
        return x

    def forward(self, x, output_shape=None):
        spatial = list(x.shape[2:])
        if output_shape is not None and list(output_shape) != spatial:
            raise NotImplementedError("resolution change is not supported by the mode-parallel layer")
        kept, _ = kept_block(spatial, self._n_modes, self.max_n_modes)
        if kept != list(self._n_modes):
            raise ValueError(f"grid {spatial} is too small for n_modes {self._n_modes} in the mode-parallel layer")
        xhat = self.ops.forward_transform(x, kept)                    # (B/P, Cin, k1, ..)
        xhat = all_to_all(xhat, split_dim=2, cat_dim=0, group=self.group)   # (B, Cin, k1/P, ..)
        yhat = self.ops.contract(xhat, self.weight)                   # (B, Cout, k1/P, ..)
        yhat = all_to_all(yhat, split_dim=0, cat_dim=2, group=self.group)   # (B/P, Cout, k1, ..)
        return self.ops.inverse_transform(yhat, self.bias, spatial)

    # ---- helpers for the training loop -----------------------------------------------------------
    def reduce_replicated_grads(self):
        """Sum the gradients of the replicated parameters (bias) over the model-parallel group
        (every rank saw a different batch shard).  The sharded weight needs nothing."""
        if self.P > 1 and self.bias is not None and self.bias.grad is not None:
            torch.distributed.all_reduce(self.bias.grad, group=self.group if self.group is not None
                                         else comm.get_model_parallel_group())

    @staticmethod
    def shard_dense_weight(full_weight, rank, world):
        """Rows of a full (Cin, Cout, k1, ..) weight that rank ``rank`` owns."""
        rows = full_weight.shape[2] // world
        return full_weight[:, :, rank * rows:(rank + 1) * rows].contiguous()
