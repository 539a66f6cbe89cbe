"""TEST ONLY: the three local stages of the spectral layer in plain torch (CPU, autograd),
built on the oracle's frequency bookkeeping.  Injected into ModeParallelSpectralConv by the
gloo tests so that the sharding / all-to-all logic can be exercised without a GPU."""
import numpy as np
import torch

from oracle import spectral_oracle as so


class OracleOps:
    def __init__(self, n_modes_attr):
        self.nm = list(n_modes_attr)

    def _index(self, spatial):
        _, freqs = so.weight_slices(list(spatial), self.nm, self.nm)
        idx = [torch.as_tensor(np.mod(f, n)) for f, n in zip(freqs[:-1], spatial[:-1])]
        idx.append(torch.as_tensor(freqs[-1]))
        return idx

    def forward_transform(self, x, kept):
        nd = x.ndim - 2
        xh = torch.fft.rfftn(x, dim=list(range(-nd, 0)), norm="forward")
        for d, ix in enumerate(self._index(x.shape[2:])):
            xh = xh.index_select(2 + d, ix)
        assert list(xh.shape[2:]) == list(kept)
        return xh

    def contract(self, xhat, w):
        return so.contract_dense(xhat, w)

    def tucker_dense(self, core, factors):
        return so.reconstruct_tucker(core, list(factors))

    def inverse_transform(self, yhat, bias, spatial):
        nd = len(spatial)
        full_shape = list(yhat.shape[:2]) + list(spatial[:-1]) + [spatial[-1] // 2 + 1]
        idx = self._index(spatial)
        cur = yhat
        # scatter dim by dim into the zero spectrum (index_add keeps autograd happy)
        for d in range(nd):
            shape = list(cur.shape)
            shape[2 + d] = full_shape[2 + d]
            z = torch.zeros(shape, dtype=cur.dtype)
            cur = z.index_add(2 + d, idx[d], cur)
        y = torch.fft.irfftn(cur, s=list(spatial), dim=list(range(-nd, 0)), norm="forward")
        return y + bias if bias is not None else y

    # one complex axis (last dim) with an explicit row -> FFT index map; norm="forward" like the layer
    def forward_axis(self, x, k, rows):
        ix = torch.as_tensor(list(rows))
        assert len(rows) == k
        return torch.fft.fft(x, dim=-1, norm="forward").index_select(-1, ix)

    def inverse_axis(self, xhat, n, rows):
        ix = torch.as_tensor(list(rows))
        z = torch.zeros(*xhat.shape[:-1], n, dtype=xhat.dtype)
        return torch.fft.ifft(z.index_add(-1, ix, xhat), dim=-1, norm="forward")


class OracleRawOps:
    """The interface of neuraloperator_amd.engine.EngineRawOps (plain calls: the stages and their adjoints, ``out=``
    slices) on the CPU: the stages are OracleOps', every adjoint is torch autograd's vector-Jacobian product of the
    corresponding (linear) stage.  Injected into mpu.ModeParallelSpectralConv by the gloo tests."""

    def __init__(self, n_modes_attr):
        self.o = OracleOps(n_modes_attr)

    @staticmethod
    def _ret(val, out):
        if out is None:
            return val
        assert out.is_contiguous() and tuple(out.shape) == tuple(val.shape)
        out.copy_(val)
        return out

    def fwd(self, x, kept):
        return self.o.forward_transform(x, kept)

    def fwd_adjoint(self, gxhat, spatial, out=None):
        n, c = gxhat.shape[:2]
        with torch.enable_grad():          # called from inside a Function's backward, where grad mode is off
            x0 = torch.zeros(n, c, *spatial, requires_grad=True)
            (gx,) = torch.autograd.grad(self.o.forward_transform(x0, list(gxhat.shape[2:])), x0, gxhat)
        return self._ret(gx, out)

    def inv(self, yhat, bias, spatial, out=None):
        b = None if bias is None else bias.reshape(-1, *(1,) * len(spatial))
        return self._ret(self.o.inverse_transform(yhat, b, spatial), out)

    def inv_adjoint(self, gy, kept, want_bias=False):
        n, c = gy.shape[:2]
        with torch.enable_grad():
            z0 = torch.zeros(n, c, *kept, dtype=torch.cfloat, requires_grad=True)
            (gh,) = torch.autograd.grad(self.o.inverse_transform(z0, None, list(gy.shape[2:])), z0, gy)
        gb = gy.sum(dim=[0] + list(range(2, gy.ndim))) if want_bias else None
        return gh, gb

    # ---- sharded layout [P][n][c][rows][rest..][2] (float32), rows past k1 zero: the all-to-all buffer in place
    @staticmethod
    def _to_shards(xh, P, rows):
        xr = torch.view_as_real(xh)
        n, c, k1 = xr.shape[:3]
        pad = xr.new_zeros((n, c, P * rows, *xr.shape[3:]))
        pad[:, :, :k1] = xr
        return pad.unflatten(2, (P, rows)).movedim(2, 0).contiguous()

    @staticmethod
    def _from_shards(buf, k1):
        return torch.view_as_complex(buf.movedim(0, 2).flatten(2, 3)[:, :, :k1].contiguous())

    def fwd_sharded(self, x, kept, P, rows, out=None, mode=None):
        return self._ret(self._to_shards(self.fwd(x, kept), P, rows), out)

    def inv_adjoint_sharded(self, gy, kept, P, rows, out=None, want_bias=False):
        gh, gb = self.inv_adjoint(gy, kept, want_bias=want_bias)
        return self._ret(self._to_shards(gh, P, rows), out), gb

    def inv_sharded(self, buf, bias, spatial, k1, out=None, mode=None):
        return self.inv(self._from_shards(buf, k1), bias, spatial, out=out)

    def fwd_adjoint_sharded(self, buf, spatial, k1, out=None):
        return self.fwd_adjoint(self._from_shards(buf, k1), spatial, out=out)

    def contract(self, xhat, w):
        return so.contract_dense(xhat, w)

    def tucker_dense(self, core, factors):
        return so.reconstruct_tucker(core, list(factors))

    def cp_dense(self, weights, factors):
        return so.reconstruct_cp(weights, list(factors))

    def tt_dense(self, cores):
        return so.reconstruct_tt(list(cores))

    def contract_separable(self, xhat, w):
        return so.contract_dense_separable(xhat, w)

    def contract_separable_bwd(self, xhat, w, ghat, need_x=True, need_w=True):
        with torch.enable_grad():
            xr, wr = xhat.detach().requires_grad_(True), w.detach().requires_grad_(True)
            gx, gw = torch.autograd.grad(so.contract_dense_separable(xr, wr), (xr, wr), ghat)
        return (gx if need_x else None), (gw if need_w else None)

    def contract_bwd(self, xhat, w, ghat, need_x=True, need_w=True):
        with torch.enable_grad():
            xr, wr = xhat.detach().requires_grad_(True), w.detach().requires_grad_(True)
            gx, gw = torch.autograd.grad(so.contract_dense(xr, wr), (xr, wr), ghat)
        return (gx if need_x else None), (gw if need_w else None)


class OracleAgOps:
    """The interface of neuraloperator_amd.engine.EngineOps (autograd stages with explicit frequency maps,
    sc_plan_desc.freq) in plain torch: what mpu.ModeParallelSpectralConv._forward_general is built from.  Default maps
    (freq None): non-last dims row r <-> signed frequency r - k // 2, last dim column c <-> c; a synthesis map entry
    None drops the row (neuraloperator_amd.modes.synthesis_freqs)."""

    @staticmethod
    def forward_transform(x, kept, freq=None):
        nd = x.ndim - 2
        spatial = list(x.shape[2:])
        xh = torch.fft.rfftn(x, dim=list(range(-nd, 0)), norm="forward")
        for d, k in enumerate(kept):
            if freq is not None and freq[d] is not None:
                ix = [int(f) for f in freq[d]]
            elif d < nd - 1:
                ix = [(r - k // 2) % spatial[d] for r in range(k)]
            else:
                ix = list(range(k))
            xh = xh.index_select(2 + d, torch.as_tensor(ix))
        return xh

    @staticmethod
    def contract(xhat, w):
        return so.contract_dense(xhat, w)

    @staticmethod
    def contract_separable(xhat, w):
        return so.contract_dense_separable(xhat, w)

    @staticmethod
    def inverse_transform(yhat, bias, spatial, freq=None, real_col=0):
        nd = len(spatial)
        kept = list(yhat.shape[2:])
        full = list(spatial[:-1]) + [spatial[-1] // 2 + 1]
        cur = yhat
        for d in range(nd):
            k = kept[d]
            if freq is not None and freq[d] is not None:
                ix = list(freq[d])
            elif d < nd - 1:
                ix = [(r - k // 2) % spatial[d] for r in range(k)]
            else:
                ix = list(range(k))
            keep = [r for r in range(k) if ix[r] is not None]
            src = cur.index_select(2 + d, torch.as_tensor(keep))
            shape = list(cur.shape)
            shape[2 + d] = full[d]
            cur = torch.zeros(shape, dtype=cur.dtype).index_add(2 + d, torch.as_tensor([int(ix[r]) for r in keep]), src)
        if real_col:                                   # spectral_convolution.py:552-556 on a resized grid
            mask = torch.ones(full[-1])
            mask[real_col] = 0.0
            cur = torch.complex(cur.real, cur.imag * mask)
        y = torch.fft.irfftn(cur, s=list(spatial), dim=list(range(-nd, 0)), norm="forward")
        return y + bias if bias is not None else y


class OracleAgOpsComplex(OracleAgOps):
    """complex_data=True: complex-to-complex transforms in every dim (spectral_convolution.py:439-441, 536-538); the
    frequency maps come from modes.analysis_freqs / synthesis_freqs (the last dim's map is always explicit)."""

    @staticmethod
    def forward_transform(x, kept, freq=None):
        nd = x.ndim - 2
        spatial = list(x.shape[2:])
        xh = torch.fft.fftn(x, dim=list(range(-nd, 0)), norm="forward")
        for d, k in enumerate(kept):
            if freq is not None and freq[d] is not None:
                ix = [int(f) for f in freq[d]]
            else:
                ix = [(r - k // 2) % spatial[d] for r in range(k)]
            xh = xh.index_select(2 + d, torch.as_tensor(ix))
        return xh

    @staticmethod
    def inverse_transform(yhat, bias, spatial, freq=None, real_col=0):
        nd = len(spatial)
        kept = list(yhat.shape[2:])
        cur = yhat
        for d in range(nd):
            k = kept[d]
            if freq is not None and freq[d] is not None:
                ix = list(freq[d])
            else:
                ix = [(r - k // 2) % spatial[d] for r in range(k)]
            keep = [r for r in range(k) if ix[r] is not None]
            src = cur.index_select(2 + d, torch.as_tensor(keep))
            shape = list(cur.shape)
            shape[2 + d] = spatial[d]
            cur = torch.zeros(shape, dtype=cur.dtype).index_add(2 + d, torch.as_tensor([int(ix[r]) for r in keep]), src)
        y = torch.fft.ifftn(cur, dim=list(range(-nd, 0)), norm="forward")
        return y + bias if bias is not None else y


class _AxisMixin:
    """one complex axis (last dim) with an explicit row -> FFT index map; norm="forward" like the layer"""

    @staticmethod
    def forward_axis(x, k, rows):
        assert len(rows) == k
        return torch.fft.fft(x, dim=-1, norm="forward").index_select(-1, torch.as_tensor(list(rows)))

    @staticmethod
    def inverse_axis(xhat, n, rows):
        z = torch.zeros(*xhat.shape[:-1], n, dtype=xhat.dtype)
        return torch.fft.ifft(z.index_add(-1, torch.as_tensor(list(rows)), xhat), dim=-1, norm="forward")


class PencilOracleOps(_AxisMixin, OracleAgOps):
    """local stages of mpu.SpatialParallelSpectralConv (round 6: frequency maps on the local transform), real data"""


class PencilOracleOpsComplex(_AxisMixin, OracleAgOpsComplex):
    """... complex_data=True"""
