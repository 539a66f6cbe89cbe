"""SphericalConv through the same plug-in (last item): the spherical-harmonic
convolution of the SFNO (/root/reference/neuralop/layers/spherical_convolution.py:206-484) on the engine's operations.

    x (B, C, nlat, nlon) --SHT--> coefficients (B, C, l, m) --contract over channels, weights W[i, o, l]--> --ISHT--> y

The reference delegates both transforms to ``torch_harmonics.RealSHT / InverseRealSHT`` (un-vendored third party, absent
here).  Their published algorithm is restated from its definition:

  SHT   X[k, m] = 2 pi rfft_lon(x)[k, m] / nlon          (longitude: real FFT, ``mmax`` columns kept)
        c[l, m] = sum_k X[k, m] w_k Pbar_l^m(cos theta_k)   (latitude: quadrature against the normalised associated
                                                           Legendre functions, Condon-Shortley phase included)
  ISHT  X[k, m] = sum_l c[l, m] Pbar_l^m(cos theta_k);  y = irfft_lon(X, n = nlon) with torch's norm="forward" (no scale)

with Clenshaw-Curtis nodes / weights on the "equiangular" grid (both poles included) and Gauss-Legendre ones on
"legendre-gauss"; ``norm`` "ortho" (orthonormal harmonics), "four-pi" and "schmidt".  Parity against torch_harmonics
itself is UNPINNED (it cannot be imported here); the tables are pinned against scipy's spherical harmonics and numpy's
Gauss-Legendre rule, the transforms by their defining properties (tests/test_spherical.py).

On the engine: the longitude transforms are 1-d plans (k_last_r2c / k_last_c2r families, sc_transform_forward /
_inverse), the two Legendre transforms and the channel contraction are sc_modegemm launches (modes = m, resp. = l),
autograd through the same Functions as the planar layer.  Constructor, attributes and ``forward`` / ``transform``
follow the reference class."""
import math
from typing import List, Optional, Union

import numpy as np
import torch
from torch import nn

from . import engine
from .factorized import DenseWeight, SpectralWeight
from .spectral_conv import BaseSpectralConv


# ------------------------------------------------------------------------------------------ tables (float64, host)
def clenshaw_curtis(n):
    """Nodes cos(theta_j), theta_j = pi j / (n - 1) (ascending from -1 to 1) and weights of the Clenshaw-Curtis rule on
    [-1, 1] (exact for polynomials of degree < n)."""
    if n < 2:
        raise ValueError("the equiangular grid needs at least 2 latitudes")
    theta = np.pi * np.arange(n) / (n - 1)
    x = np.cos(theta)[::-1].copy()
    w = np.zeros(n)
    N = n - 1
    jj = np.arange(1, N // 2 + 1)
    for i in range(n):
        b = np.where(2 * jj == N, 1.0, 2.0)
        s = np.sum(b / (4.0 * jj * jj - 1.0) * np.cos(2.0 * jj * theta[i]))
        c = 1.0 if i in (0, N) else 2.0
        w[i] = c / N * (1.0 - s)
    return x, w[::-1].copy()


def quadrature(nlat, grid):
    """(colatitudes theta ascending from the north pole, weights in cos theta)."""
    if grid == "equiangular":
        x, w = clenshaw_curtis(nlat)
    elif grid == "legendre-gauss":
        x, w = np.polynomial.legendre.leggauss(nlat)
    else:
        raise ValueError(f"grid {grid!r}: 'equiangular' or 'legendre-gauss'")
    theta = np.arccos(np.clip(x, -1.0, 1.0))[::-1].copy()
    return theta, w[::-1].copy()


def legendre_table(mmax, lmax, theta, norm="ortho", inverse=False):
    """Pbar[m, l, k] = normalised associated Legendre function of degree l, order m at cos(theta_k), Condon-Shortley
    phase included, zero for l < m.  Stable three-term recurrence on the orthonormal functions."""
    ct, st = np.cos(theta), np.sin(theta)
    nmax = max(mmax, lmax)
    p = np.zeros((nmax, nmax, len(theta)))
    p[0, 0] = math.sqrt(1.0 / (4.0 * math.pi))
    for m in range(1, nmax):
        p[m, m] = -math.sqrt((2.0 * m + 1.0) / (2.0 * m)) * st * p[m - 1, m - 1]
    for m in range(nmax - 1):
        p[m, m + 1] = math.sqrt(2.0 * m + 3.0) * ct * p[m, m]
    for m in range(nmax):
        for l in range(m + 2, nmax):
            a = math.sqrt((4.0 * l * l - 1.0) / (l * l - m * m))
            b = math.sqrt(((l - 1.0) ** 2 - m * m) / (4.0 * (l - 1.0) ** 2 - 1.0))
            p[m, l] = a * (ct * p[m, l - 1] - b * p[m, l - 2])
    p = p[:mmax, :lmax]
    if norm == "ortho":
        f = np.ones(lmax)
    elif norm == "four-pi":
        f = np.full(lmax, math.sqrt(4.0 * math.pi))
    elif norm == "schmidt":
        f = np.sqrt(4.0 * math.pi / (2.0 * np.arange(lmax) + 1.0))
    else:
        raise ValueError(f"norm {norm!r}: 'ortho', 'four-pi' or 'schmidt'")
    if inverse:
        f = 1.0 / f
    return p * f[None, :, None]


class SHT(nn.Module):
    """Spherical-harmonic transforms with the call interface of the reference's wrapper (:206-281), on the engine."""

    def __init__(self, dtype=torch.float32, device=None, engine_flags=0):
        super().__init__()
        self.device, self.dtype, self.flags = device, dtype, engine_flags
        self._tab = {}

    def _table(self, key, build, dev):
        k = (key, str(dev))
        if k not in self._tab:
            t = torch.from_numpy(np.ascontiguousarray(build())).to(torch.float32)
            self._tab[k] = torch.complex(t, torch.zeros_like(t)).to(dev).contiguous()
        return self._tab[k]

    def sht(self, x, s=None, norm="ortho", grid="equiangular"):
        *lead, nlat, nlon = x.shape
        lmax, mmax = (nlat, nlat // 2 if grid == "equiangular" else nlat) if s is None else (int(s[0]), int(s[1]))
        if mmax > nlon // 2 + 1:
            raise ValueError(f"mmax = {mmax} exceeds the {nlon // 2 + 1} longitudinal modes of {nlon} points")

        def build():
            theta, w = quadrature(nlat, grid)
            return (legendre_table(mmax, lmax, theta, norm) * w[None, None, :]).transpose(2, 1, 0)   # [k, l, m]

        wt = self._table(("f", nlat, lmax, mmax, norm, grid), build, x.device)
        ops = engine.EngineOps("forward", self.flags)
        n_lines = 1
        for v in lead:
            n_lines *= int(v)
        xh = ops.forward_transform(x.reshape(1, n_lines * nlat, nlon), [mmax])           # rfft / nlon, mmax columns
        xh = xh.reshape(n_lines, nlat, mmax) * (2.0 * math.pi)
        out = engine.mode_gemm(xh, wt, mmax)                                              # [lines, lmax, mmax]
        return out.reshape(*lead, lmax, mmax)

    def isht(self, x, s=None, norm="ortho", grid="equiangular"):
        *lead, lmax, mmax = x.shape
        nlat, nlon = (lmax, 2 * mmax if grid == "equiangular" else mmax) if s is None else (int(s[0]), int(s[1]))

        def build():
            theta, _ = quadrature(nlat, grid)
            return legendre_table(mmax, lmax, theta, norm, inverse=True).transpose(1, 2, 0)            # [l, k, m]

        pt = self._table(("i", nlat, lmax, mmax, norm, grid), build, x.device)
        n_lines = 1
        for v in lead:
            n_lines *= int(v)
        xh = engine.mode_gemm(x.reshape(n_lines, lmax, mmax).to(torch.complex64), pt, mmax)   # [lines, nlat, mmax]
        keep = min(mmax, nlon // 2 + 1)                      # irfft(n = nlon) reads the first nlon / 2 + 1 columns only
        if keep < mmax:
            xh = xh[..., :keep].contiguous()
        ops = engine.EngineOps("forward", self.flags)
        y = ops.inverse_transform(xh.reshape(1, n_lines * nlat, keep), None, [nlon])
        return y.reshape(*lead, nlat, nlon)


class SphericalConv(BaseSpectralConv):
    """Drop-in for neuralop.layers.spherical_convolution.SphericalConv (:284-484): the ``conv_module`` of the SFNO
    (models/sfno.py:7-9).  Weights (in, out, l) -- the contraction is diagonal in l and m and does not depend on m
    ("dhconv") -- as dense / Tucker / CP / TT containers (contracted through their dense tensor)."""

    def __init__(self, in_channels, out_channels, n_modes, max_n_modes=None, bias=True, separable=False,
                 resolution_scaling_factor: Optional[Union[float, List[float]]] = None, fno_block_precision="full",
                 rank=0.5, factorization="cp", implementation="reconstructed", fixed_rank_modes=False,
                 joint_factorization=False, decomposition_kwargs=None, init_std="auto", sht_norm="ortho",
                 sht_grids="equiangular", device=None, dtype=torch.float32, complex_data=False, engine_flags=0):
        super().__init__(dtype=dtype, device=device)
        self.in_channels, self.out_channels = in_channels, out_channels
        self.joint_factorization = joint_factorization
        n_modes = [n_modes] if isinstance(n_modes, int) else list(n_modes)
        self._n_modes = n_modes
        self.order = len(n_modes)
        self.max_n_modes = list(n_modes) if max_n_modes is None else \
            ([max_n_modes] if isinstance(max_n_modes, int) else list(max_n_modes))
        self.rank, self.factorization, self.implementation = rank, factorization, implementation
        if resolution_scaling_factor is not None and isinstance(resolution_scaling_factor, (int, float)):
            resolution_scaling_factor = [float(resolution_scaling_factor)] * self.order
        self.resolution_scaling_factor = resolution_scaling_factor
        if init_std == "auto":
            init_std = (2 / (in_channels + out_channels)) ** 0.5
        if isinstance(fixed_rank_modes, bool):
            fixed_rank_modes = [0] if fixed_rank_modes else None
        fac = "Dense" if factorization is None else factorization
        if fac.lower().startswith("complex"):
            fac = fac[len("complex"):]
        if separable:
            if in_channels != out_channels:
                raise ValueError("To use separable Fourier Conv, in_channels must be equal to out_channels, but got "
                                 f"in_channels={in_channels} and out_channels={out_channels}")
            weight_shape = (in_channels, *self.n_modes[:-1])
        else:
            weight_shape = (in_channels, out_channels, *self.n_modes[:-1])
        self.separable = separable
        self.weight = SpectralWeight.new(weight_shape, rank=rank, factorization=fac, fixed_rank_modes=fixed_rank_modes,
                                         dtype=torch.cfloat, device=device, **(decomposition_kwargs or {}))
        self.weight.normal_(0, init_std)
        self.bias = nn.Parameter(init_std * torch.randn(out_channels, *(1,) * self.order, device=device)) if bias else None
        self.sht_norm = sht_norm
        self.sht_grids = [sht_grids] * 2 if isinstance(sht_grids, str) else list(sht_grids)
        self.sht_handle = SHT(dtype=dtype, device=device, engine_flags=engine_flags)

    @property
    def n_modes(self):
        return self._n_modes

    @n_modes.setter
    def n_modes(self, n_modes):
        self._n_modes = [n_modes] if isinstance(n_modes, int) else list(n_modes)

    def _out_size(self, in_height, in_width, output_shape):
        if self.resolution_scaling_factor is not None and output_shape is None:
            return round(in_height * self.resolution_scaling_factor[0]), round(in_width * self.resolution_scaling_factor[1])
        if output_shape is not None:
            return int(output_shape[0]), int(output_shape[1])
        return in_height, in_width

    def transform(self, x, output_shape=None):                                            # :408-428
        *_, in_height, in_width = x.shape
        height, width = self._out_size(in_height, in_width, output_shape)
        if (in_height, in_width) == (height, width) and self.sht_grids[0] == self.sht_grids[1]:
            return x
        coefs = self.sht_handle.sht(x, s=self.n_modes, norm=self.sht_norm, grid=self.sht_grids[0])
        return self.sht_handle.isht(coefs, s=(height, width), norm=self.sht_norm, grid=self.sht_grids[1])

    def _dense_weight(self):
        w = self.weight
        return w.tensor if isinstance(w, DenseWeight) else w.to_tensor()

    def forward(self, x, output_shape=None):                                              # :430-472
        b, _, in_height, in_width = x.shape
        height, width = self._out_size(in_height, in_width, output_shape)
        L, M = int(self.n_modes[0]), int(self.n_modes[1]) // 2
        c = self.sht_handle.sht(x.float(), s=(L, M), norm=self.sht_norm, grid=self.sht_grids[0])   # (B, Ci, L, M)
        w = self._dense_weight()[..., :L]
        ci = int(c.shape[1])
        # out[b, o, l, m] = sum_i c[b, i, l, m] W[i, o, l]: modes = l, the m axis rides with the batch
        a = c.permute(0, 3, 1, 2).reshape(b * M, ci, L)
        if self.separable:
            y = (a * w.unsqueeze(0)).reshape(b, M, ci, L)
        else:
            y = engine.mode_gemm(a, w, L).reshape(b, M, int(w.shape[1]), L)
        y = y.permute(0, 2, 3, 1).contiguous()                                            # (B, Co, L, M)
        out = self.sht_handle.isht(y, s=(height, width), norm=self.sht_norm, grid=self.sht_grids[1])
        return out + self.bias if self.bias is not None else out
