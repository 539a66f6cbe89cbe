"""Multigrid patching on this package's collectives: the domain decomposition the
reference wraps around a model for large grids (/root/reference/neuralop/training/patching.py:13-376).

Same class name, constructor and methods (``patch`` / ``unpatch``; module-level ``make_patches``), same tensor layouts:

* ``make_patches(x, n, p)``: every sample is cut into ``n1 x n2`` equal tiles, each with a ``p``-pixel halo taken
  periodically from its neighbours, stacked along the batch dim (sample-major, then tile row, then tile column);
* ``MultigridPatching2D._make_mg_patches``: on top of that, for every level ``l = 1..levels`` a view of the WHOLE
  field subsampled by ``2**l`` and windowed around the same tile (same window size, stride tile / 2**l, periodic),
  concatenated along the channel dim -- the coarse context the model sees next to its fine patch;
* ``patch`` scatters the stacked patches (and the un-stitched targets) over the model-parallel group along the batch
  dim, ``unpatch`` removes the halo, gathers and stitches.

Formulated as ONE periodic gather per level (index arithmetic modulo the grid) instead of the reference's circular
``pad`` + ``unfold`` + ``permute`` chains: no padded copies, and halos wider than the field need no special case.
"""
import math

import torch
from torch import nn

from . import comm
from .mappings import gather_from_model_parallel_region, scatter_to_model_parallel_region


def _pair(v):
    return [v, v] if isinstance(v, (int, float)) else list(v)


def _window_gather(x, starts, win, stride=None):
    """x (B, C, H, W) -> (B * n1 * n2, C, win1, win2): window (i1, i2) of every sample starts at
    (starts[0] + i1 * stride[0], starts[1] + i2 * stride[1]) and reads ``win`` pixels per dim, periodically."""
    b, c, h, w = x.shape
    (o1, n1, s1), (o2, n2, s2) = starts
    rows = (o1 + s1 * torch.arange(n1, device=x.device)[:, None] + torch.arange(win[0], device=x.device)[None, :]) % h
    cols = (o2 + s2 * torch.arange(n2, device=x.device)[:, None] + torch.arange(win[1], device=x.device)[None, :]) % w
    t = x[:, :, rows.reshape(-1), :][:, :, :, cols.reshape(-1)]                  # (B, C, n1 win1, n2 win2)
    t = t.reshape(b, c, n1, win[0], n2, win[1]).permute(0, 2, 4, 1, 3, 5)        # (B, n1, n2, C, win1, win2)
    return t.reshape(b * n1 * n2, c, win[0], win[1])


def make_patches(x, n, p=0):
    """(B, C, H, W) -> (B n1 n2, C, H / n1 + 2 p1, W / n2 + 2 p2); 1-D: (B, C, S) -> (B n, C, S / n + 2 p)
    (patching.py:304-376)."""
    if x.ndim not in (3, 4):
        raise ValueError(f"make_patches takes (B, C, S) or (B, C, H, W) tensors, got {x.ndim} dims")
    if x.ndim == 3:
        n1 = n if isinstance(n, int) else n[0]
        p1 = p if isinstance(p, int) else p[0]
        return make_patches(x.unsqueeze(2), [1, n1], [0, p1]).squeeze(2)
    n, p = _pair(n), _pair(p)
    h, w = x.shape[-2:]
    if n[0] <= 1 and n[1] <= 1:                       # the reference returns the padded field itself
        return _window_gather(x, ((-p[0], 1, 0), (-p[1], 1, 0)), (h + 2 * p[0], w + 2 * p[1])) if (p[0] or p[1]) else x
    if h % n[0] or w % n[1]:
        raise ValueError(f"grid {h} x {w} does not split into {n[0]} x {n[1]} equal patches")
    th, tw = h // n[0], w // n[1]
    return _window_gather(x, ((-p[0], n[0], th), (-p[1], n[1], tw)), (th + 2 * p[0], tw + 2 * p[1]))


class MultigridPatching2D(nn.Module):
    def __init__(self, model, levels=0, padding_fraction=0, use_distributed=False, stitching=True):
        super().__init__()
        self.levels = levels
        self.padding_fraction = _pair(padding_fraction)
        self.n_patches = [2 ** levels, 2 ** levels]
        self.model = model
        self.use_distributed = use_distributed
        self.stitching = stitching
        self.padding_height = self.padding_width = 0
        if levels > 0:
            print(f"MGPatching(n_patches={self.n_patches}, padding_fraction={self.padding_fraction}, "
                  f"levels={self.levels}, use_distributed={use_distributed}, stitching={stitching})")
        # every rank back-propagates the STITCHED field: undo the averaging of the data-parallel reduction
        # (patching.py:76-81)
        if self.use_distributed and self.stitching:
            for param in model.parameters():
                param.register_hook(lambda grad: grad * float(comm.get_model_parallel_size()))

    # ---- patching.py:83-106
    def patch(self, x, y):
        if not self.stitching:
            y = make_patches(y, n=self.n_patches, p=0)
        if self.use_distributed:
            y = scatter_to_model_parallel_region(y, 0)
        x = self._make_mg_patches(x)
        if self.use_distributed:
            x = scatter_to_model_parallel_region(x, 0)
        return x, y

    # ---- patching.py:108-145
    def unpatch(self, x, y, evaluation=False):
        if self.padding_height > 0 or self.padding_width > 0:
            x = self._unpad(x)
        if self.use_distributed and self.stitching:
            x = gather_from_model_parallel_region(x, dim=0)
        if self.stitching or evaluation:
            x = self._stitch(x)
        if evaluation and not self.stitching:
            y = self._stitch(y)
        return x, y

    # ---- patching.py:147-190: (B n1 n2, C, h, w) -> (B, C, n1 h, n2 w)
    def _stitch(self, x):
        if x.ndim != 4:
            raise ValueError(f"Only 2D patch supported but got input with {x.ndim} dims.")
        n1, n2 = self.n_patches
        if n1 <= 1 and n2 <= 1:
            return x
        bn, c, h, w = x.shape
        b = bn // (n1 * n2)
        return x.reshape(b, n1, n2, c, h, w).permute(0, 3, 1, 4, 2, 5).reshape(b, c, n1 * h, n2 * w)

    # ---- patching.py:192-283
    def _make_mg_patches(self, x):
        levels = self.levels
        if levels <= 0:
            return x
        _, _, height, width = x.shape
        pad = [int(round(height * self.padding_fraction[0])), int(round(width * self.padding_fraction[1]))]
        self.padding_height, self.padding_width = pad
        n = 2 ** levels
        parts = [make_patches(x, n=n, p=pad)]
        tile = [parts[0].size(-2) - 2 * pad[0], parts[0].size(-1) - 2 * pad[1]]
        win = (tile[0] + 2 * pad[0], tile[1] + 2 * pad[1])
        for level in range(1, levels + 1):
            sub = 2 ** level
            xs = x[:, :, ::sub, ::sub]
            starts = []
            for d in range(2):
                stride = tile[d] // sub
                # the coarse windows are centred on the same tiles: total span of the n windows minus the coarse
                # field, split evenly (rounded up) to the left, plus the halo (patching.py:232-245)
                lead = math.ceil((tile[d] + (n - 1) * stride - xs.size(-2 + d)) / 2.0) + pad[d]
                span = xs.size(-2 + d) + 2 * lead
                count = (span - win[d]) // stride + 1 if stride > 0 else 0
                if count != n:
                    raise ValueError(f"grid {height} x {width} with levels={levels}, padding {pad}: level {level} "
                                     f"yields {count} coarse windows per dim instead of {n}")
                starts.append((-lead, n, stride))
            parts.append(_window_gather(xs, tuple(starts), win))
        return torch.cat(parts, dim=1)

    # ---- patching.py:285-301
    def _unpad(self, x):
        ph, pw = self.padding_height, self.padding_width
        return x[..., ph:x.size(-2) - ph, pw:x.size(-1) - pw].contiguous()
