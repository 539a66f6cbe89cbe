"""Drop-in ``conv_module`` for the reference's FNOBlocks / FNO / TFNO.

Mirrors the constructor, attributes and methods of
/root/reference/neuralop/layers/spectral_convolution.py:183-570 (``SpectralConv``) and the
plug-in contract of neuralop/layers/base_spectral_conv.py:4-27 -- ``forward(x,
output_shape=None)``, ``transform(x, output_shape=None)``, mutable ``n_modes``,
``max_n_modes``, ``weight``, ``bias`` -- so that ``FNO(..., conv_module=SpectralConv)``
(fno_block.py:210-240) runs the MI355X engine for every Fourier layer.

All arithmetic of the layer is in libsc_engine.so; this file is argument checking and
autograd plumbing.  Dense / Tucker / CP / TT weights, separable weights, complex data and
resolution-changing layers (``resolution_scaling_factor`` / ``output_shape``, with the reference's
end-padding behaviour) all run on the engine -- there is no silent PyTorch fallback.
``fno_block_precision="half"/"mixed"`` (:436-459): values are rounded to float16 at the reference's cast points and
the contraction follows ``einsum_complexhalf`` (einsum_utils.py:10-36) -- pinned against the verbatim function; the two
transforms compute in fp32 (the reference's float16 FFT has no CPU backend to pin against).  The result has the dtype
the reference returns (fp32 with a bias, fp16 without).
"""
from typing import List, Optional, Tuple, Union

import torch
from torch import nn

from . import _lib, engine
from .factorized import CPWeight, DenseWeight, SpectralWeight, TTWeight, TuckerWeight
from . import modes
from .modes import halve_last_mode, kept_block

Number = Union[int, float]


def _validate_scaling_factor(scaling_factor, n_dim):
    """Single-layer form of neuralop/utils.py:151-197."""
    if scaling_factor is None:
        return None
    if isinstance(scaling_factor, (float, int)):
        return [float(scaling_factor)] * n_dim
    if isinstance(scaling_factor, (list, tuple)) and len(scaling_factor) == n_dim and all(
            isinstance(s, (float, int)) for s in scaling_factor):
        return [float(s) for s in scaling_factor]
    raise ValueError(f"resolution_scaling_factor={scaling_factor!r} not understood for {n_dim}-d")


class BaseSpectralConv(nn.Module):
    """Same contract as neuralop/layers/base_spectral_conv.py:4-27."""

    def __init__(self, device=None, dtype=None):
        super().__init__()
        self.dtype = dtype
        self.device = device

    def transform(self, x):
        return x


class SpectralConv(BaseSpectralConv):
    """N-d Fourier layer on the MI355X engine (real-valued data, fp32 spectral arithmetic).

    Parameters are those of the reference class (spectral_convolution.py:285-305); see the
    module docstring for the variants that raise NotImplementedError.
    """

    def __init__(
        self,
        in_channels,
        out_channels,
        n_modes,
        complex_data=False,
        max_n_modes=None,
        bias=True,
        separable=False,
        resolution_scaling_factor: Optional[Union[Number, List[Number]]] = None,
        fno_block_precision="full",
        rank=1.0,
        factorization=None,
        implementation="reconstructed",
        enforce_hermitian_symmetry=True,
        fixed_rank_modes=False,
        decomposition_kwargs: Optional[dict] = None,
        init_std="auto",
        fft_norm="forward",
        device=None,
        engine_flags: int = 0,
    ):
        super().__init__(device=device)
        if fno_block_precision not in ("full", "half", "mixed"):
            raise ValueError(f"fno_block_precision={fno_block_precision!r}: expected full, half or mixed")
        if implementation not in ("reconstructed", "factorized"):
            raise ValueError(
                f'Got implementation={implementation}, expected "reconstructed" or "factorized"')
        if fft_norm not in ("forward", "backward", "ortho"):
            raise ValueError(f"fft_norm={fft_norm!r}")
        self.in_channels = in_channels
        self.out_channels = out_channels
        self.complex_data = complex_data
        self.n_modes = n_modes
        self.order = len(self.n_modes)
        if self.order > 4:
            raise NotImplementedError("the engine supports 1-d .. 4-d grids")

        if max_n_modes is None:
            max_n_modes = self.n_modes
        elif isinstance(max_n_modes, int):
            max_n_modes = [max_n_modes]
        self.max_n_modes = max_n_modes

        self.fno_block_precision = fno_block_precision
        self.rank = rank
        self.factorization = factorization
        self.implementation = implementation
        # the engine's C2R ignores Im of the DC / Nyquist columns by construction, which is what
        # the reference's explicit fix-up enforces (:547-559); both settings give the same result
        self.enforce_hermitian_symmetry = enforce_hermitian_symmetry
        self.resolution_scaling_factor = _validate_scaling_factor(resolution_scaling_factor, self.order)
        self.engine_flags = engine_flags

        if init_std == "auto":
            init_std = (2 / (in_channels + out_channels)) ** 0.5
        if isinstance(fixed_rank_modes, bool):
            fixed_rank_modes = [0] if fixed_rank_modes else None
        self.fft_norm = fft_norm

        if separable:
            if in_channels != out_channels:
                raise ValueError(
                    "To use separable Fourier Conv, in_channels must be equal "
                    f"to out_channels, but got in_channels={in_channels} and "
                    f"out_channels={out_channels}",
                )
            weight_shape = (in_channels, *self.max_n_modes)                # :347-353
        else:
            weight_shape = (in_channels, out_channels, *self.max_n_modes)
        self.separable = separable

        tensor_kwargs = decomposition_kwargs if decomposition_kwargs is not None else {}
        self.weight = SpectralWeight.new(
            weight_shape, rank=self.rank, factorization=factorization or "Dense",
            fixed_rank_modes=fixed_rank_modes, dtype=torch.cfloat, device=device, **tensor_kwargs)
        self.weight.normal_(0, init_std)

        if bias:
            self.bias = nn.Parameter(
                init_std * torch.randn(*(tuple([self.out_channels]) + (1,) * self.order), device=device))
        else:
            self.bias = None

    # ---- plug-in contract -------------------------------------------------------------------
    def transform(self, x, output_shape=None):
        in_shape = list(x.shape[2:])
        if self.resolution_scaling_factor is not None and output_shape is None:
            out_shape = [round(s * r) for (s, r) in zip(in_shape, self.resolution_scaling_factor)]
        elif output_shape is not None:
            out_shape = list(output_shape)
        else:
            out_shape = in_shape
        if in_shape == out_shape:
            return x
        return self._resample(x, out_shape)

    def _resample(self, x, out_shape):
        """``resample(x, 1.0, spatial dims, output_shape)`` of neuralop/layers/resample.py:7-71: 1-d linear and
        2-d bicubic interpolation are ATen's interpolators exactly as in the reference (:49-52); 3-d and
        up is the spectral resample (:54-66) on the engine -- truncated forward transform on the old grid,
        zero-padded inverse on the new one, rows by the resample's own convention (modes.resample_block)."""
        nd = x.ndim - 2
        out_shape = [int(v) for v in out_shape]
        if nd == 1:
            return torch.nn.functional.interpolate(x, size=out_shape[0], mode="linear", align_corners=True)
        if nd == 2:
            return torch.nn.functional.interpolate(x, size=tuple(out_shape), mode="bicubic", align_corners=True)
        kept, fa, fs = modes.resample_block(list(x.shape[2:]), out_shape)
        ops = engine.EngineOps("forward", self.engine_flags)
        return ops.inverse_transform(ops.forward_transform(x.float(), kept, fa), None, out_shape, fs)

    @property
    def n_modes(self):
        return self._n_modes

    @n_modes.setter
    def n_modes(self, n_modes):
        self._n_modes = halve_last_mode(n_modes, complex_data=self.complex_data)

    # ---- forward ------------------------------------------------------------------------------
    def _dense_weight(self):
        w = self.weight
        if isinstance(w, DenseWeight):
            return w.tensor
        # Tucker / CP: rebuild the dense weight from the (small) factors on the device, then run
        # the dense engine path -- the identity the reference's own test pins
        # (neuralop/layers/tests/test_spectral_convolution.py:54-65).
        return w.to_tensor()

    def forward(self, x: torch.Tensor, output_shape: Optional[Tuple[int]] = None):
        if x.ndim != self.order + 2:
            raise ValueError(f"expected a (B, C, {self.order} spatial dims) input, got {tuple(x.shape)}")
        spatial = list(x.shape[2:])
        out_shape = spatial
        if self.resolution_scaling_factor is not None and output_shape is None:
            out_shape = [round(s * r) for (s, r) in zip(spatial, self.resolution_scaling_factor)]
        if output_shape is not None:
            out_shape = list(output_shape)
        out_shape = [int(v) for v in out_shape]
        if self.fno_block_precision in ("half", "mixed") and not self.complex_data:
            if x.is_cuda and not self.separable and out_shape == spatial:
                return self._forward_half(x.float(), spatial)
            y = self._forward_full(x.float(), spatial, out_shape)    # separable / resized: fp32 arithmetic
            return y if self.bias is not None else y.half()      # half + fp32 bias promotes to fp32 upstream
        if x.dtype == torch.bfloat16:
            # bfloat16 activations (BASELINE configs[1] "bf16"; torch.fft has no bfloat16, so upstream has no
            # behaviour to match): y = bf16(layer(fp32(x))).  On the fused 2-D kernels x / y / their gradients
            # cross HBM as bfloat16 (SC_PLAN_IO_BF16); every other route converts around the fp32 engine.
            return self._forward_full(x, spatial, out_shape).to(torch.bfloat16)
        return self._forward_full(x, spatial, out_shape)

    def forward_fused(self, x, skip, activation="gelu"):
        """``activation(self(x) + skip)`` -- the Fourier layer of an FNO block (fno_block.py:392-414) -- with the
        addition and the activation inside the inverse transform's store path (three R-sized
        elementwise passes less per layer).  Dense weights, real data, unchanged grid; every other configuration
        takes the unfused composition of the same operations."""
        spatial = list(x.shape[2:])
        dense = isinstance(self.weight, DenseWeight) and not self.separable and not self.complex_data
        if dense and self.resolution_scaling_factor is None and x.dtype == torch.float32 and \
                self.fno_block_precision == "full":
            return engine.FourierLayerFn.apply(x, self._dense_weight(), self.bias, skip, activation,
                                               list(self.n_modes), list(self.max_n_modes), self.fft_norm,
                                               self.engine_flags)
        y = self(x) + skip
        return torch.nn.functional.gelu(y) if activation == "gelu" else y

    def _forward_half(self, x, spatial):
        """``fno_block_precision`` "half" / "mixed" (spectral_convolution.py:436-459): the reference casts x to
        float16 ("half"), the spectrum to complex32 (both) and contracts with ``einsum_complexhalf``
        (einsum_utils.py:10-36); the inverse transform of the complex32 spectrum returns float16.  Here every one of
        those values is ROUNDED to float16 at the same point and kept in fp32 storage: the contraction reproduces the
        reference's four-real-product arithmetic (SC_GEMM_F16, pinned against the verbatim einsum on the CPU), the two
        transforms compute in fp32 (the reference's float16 FFT arithmetic is the backend's; it has no CPU
        implementation to pin against).  Factorized weights contract through their dense block."""
        kept, wsl = self._used_block(spatial)
        w = self._block_dense(wsl, kept)
        ops = engine.EngineOps(self.fft_norm, self.engine_flags)
        if self.fno_block_precision == "half":
            x = engine.round_f16(x)                                                   # :436-437
        xhat = ops.forward_transform(x, kept)
        b, ci = int(xhat.shape[0]), int(xhat.shape[1])
        co, m = int(w.shape[1]), 1
        for k in kept:
            m *= int(k)
        yhat = engine.mode_gemm(xhat.reshape(b, ci, m), w.reshape(ci, co, m), m, flags=_lib.SC_GEMM_F16)
        y = engine.round_f16(ops.inverse_transform(yhat.reshape(b, co, *kept), None, spatial))
        return y + self.bias if self.bias is not None else y.half()

    def _forward_full(self, x, spatial, out_shape):
        if self.complex_data or out_shape != spatial:
            return self._forward_staged(x, spatial, out_shape)
        if self.separable:
            return self._forward_separable(x, spatial)
        if isinstance(self.weight, TTWeight) and x.is_cuda:
            kept, wsl = self._used_block(spatial)
            return engine.SpectralConvDenseFn.apply(x, self._tt_dense(wsl, kept), self.bias, list(kept),
                                                    list(kept), self.fft_norm, self.engine_flags)
        if isinstance(self.weight, CPWeight) and x.is_cuda:
            if self.implementation == "factorized":
                return self._forward_cp(x, spatial)
            kept, wsl = self._used_block(spatial)
            return engine.SpectralConvDenseFn.apply(x, self._cp_dense(wsl, kept), self.bias, list(kept),
                                                    list(kept), self.fft_norm, self.engine_flags)
        if isinstance(self.weight, TuckerWeight) and x.is_cuda:
            if self.implementation == "factorized":
                return self._forward_tucker(x, spatial)
            kept, wsl = self._used_block(spatial)           # "reconstructed": rebuild only the used block
            return engine.SpectralConvDenseFn.apply(x, self._tucker_dense(wsl, kept), self.bias, list(kept),
                                                    list(kept), self.fft_norm, self.engine_flags)
        return engine.SpectralConvDenseFn.apply(
            x, self._dense_weight(), self.bias, list(self.n_modes), list(self.max_n_modes),
            self.fft_norm, self.engine_flags)

    # ---- Tucker weights on the engine -----------------------------------------------------------
    def _used_block(self, spatial):
        kept, w_start = kept_block(spatial, list(self.n_modes), list(self.max_n_modes))
        if not any(w_start) and list(kept) == [int(v) for v in self.max_n_modes]:
            return kept, self.weight                        # the whole stored block: nothing to slice (host time)
        lead = (slice(None),) if self.separable else (slice(None), slice(None))       # :471-474
        idx = lead + tuple(slice(s0, s0 + k) for s0, k in zip(w_start, kept))
        return kept, self.weight[idx]                       # factors row-sliced to the used block

    # ---- complex data / a different output grid: stage by stage with two plans --------------------
    def _block_dense(self, wsl, kept):
        """dense (Cin, Cout, *kept) -- (C, *kept) when separable -- tensor of the used weight block"""
        if torch.is_tensor(wsl):
            return wsl
        if not self.separable and isinstance(wsl, TuckerWeight):
            return self._tucker_dense(wsl, kept)
        if not self.separable and isinstance(wsl, TTWeight):
            return self._tt_dense(wsl, kept)
        if not self.separable and isinstance(wsl, CPWeight):
            return self._cp_dense(wsl, kept)
        return wsl.to_tensor()

    def _forward_staged(self, x, spatial, out_shape):
        """Forward transform on the input grid, contraction, inverse transform onto ``out_shape`` with the
        reference's placement of the kept rows (modes.synthesis_freqs: rows keep their input-grid FFT
        index, spectral_convolution.py:524-559), each its own plan; complex data (:439-441, 536-538)
        takes the same route with complex-to-complex passes in every dim."""
        cplx = self.complex_data
        if cplx:
            kept, w_start = modes.kept_block_complex(spatial, list(self.n_modes), list(self.max_n_modes))
        else:
            kept, w_start = kept_block(spatial, list(self.n_modes), list(self.max_n_modes))
        lead = (slice(None),) if self.separable else (slice(None), slice(None))
        wsl = self.weight[lead + tuple(slice(s0, s0 + k) for s0, k in zip(w_start, kept))]
        w = self._block_dense(wsl, kept)
        fa = modes.analysis_freqs(spatial, kept, cplx)
        fs, real_col = modes.synthesis_freqs(spatial, out_shape, kept, cplx)
        ops = engine.EngineOps(self.fft_norm, self.engine_flags | (engine.SC_PLAN_COMPLEX if cplx else 0))
        xhat = ops.forward_transform(x, kept, fa)
        if self.separable:
            b, c = xhat.shape[:2]
            m = c
            for k in kept:
                m *= int(k)
            yhat = engine.mode_gemm(xhat.reshape(b, 1, m), w.reshape(1, 1, m), m).reshape(b, c, *kept)
        else:
            yhat = ops.contract(xhat, w)
        if cplx:                          # real bias added to a complex field: elementwise glue (:567-568)
            y = ops.inverse_transform(yhat, None, out_shape, fs)
            return y if self.bias is None else y + self.bias
        return ops.inverse_transform(yhat, self.bias, out_shape, fs, real_col)

    # ---- depth-wise (separable) weights -----------------------------------------------------------
    def _forward_separable(self, x, spatial):
        """yhat[b,c,m] = xhat[b,c,m] * w[c,m] (``_contract_dense_separable``, spectral_convolution.py:49-52;
        factorized separable weights are rebuilt first, the identity test_spectral_convolution.py:54-65
        pins).  One sc_modegemm launch with (channel, mode) as its lane index and R = Q = 1; its
        autograd is ModeGemmFn's."""
        kept, wsl = self._used_block(spatial)
        w = wsl if torch.is_tensor(wsl) else wsl.to_tensor()                       # (C, *kept)
        ops = engine.EngineOps(self.fft_norm, self.engine_flags)
        xhat = ops.forward_transform(x, kept)                                      # (B, C, *kept)
        b, c = xhat.shape[:2]
        m = c
        for k in kept:
            m *= int(k)
        yhat = engine.mode_gemm(xhat.reshape(b, 1, m), w.reshape(1, 1, m), m)
        return ops.inverse_transform(yhat.reshape(b, c, *kept), self.bias, spatial)

    # ---- CP weights on the engine ------------------------------------------------------------------
    @staticmethod
    def _cp_mode_rows(wsl, kept):
        """S[r, modes] = lambda_r * prod_d U_d[m_d, r]: the Khatri-Rao row products of the (small) mode
        factors -- parameter-space elementwise glue, R x modes numbers."""
        s = wsl.weights
        for u in wsl.factors[2:]:
            s = s.unsqueeze(-1) * u.transpose(0, 1).reshape(u.shape[1], *([1] * (s.dim() - 1)), u.shape[0])
        m = 1
        for k in kept:
            m *= int(k)
        return s.reshape(s.shape[0], m)

    def _cp_dense(self, wsl, kept):
        """W[(i,o), m] = sum_r (U_in[i,r] U_out[o,r]) S[r,m]: one sc_modegemm launch (lanes = modes)."""
        u_in, u_out = wsl.factors[0], wsl.factors[1]
        s = self._cp_mode_rows(wsl, kept)
        r, m = int(s.shape[0]), int(s.shape[1])
        ab = (u_in.unsqueeze(1) * u_out.unsqueeze(0)).reshape(-1, r)               # (Cin Cout, R)
        w = engine.mode_gemm(ab, s.reshape(r, 1, m), m)
        return w.reshape(u_in.shape[0], u_out.shape[0], *kept)

    def _forward_cp(self, x, spatial):
        """implementation="factorized" with a CP weight, 'abcd,r,br,er,cr,dr->aecd'
        (spectral_convolution.py:55-73) in its minimum-FLOP pairwise order, never forming the dense weight:
            z[b,r,m] = sum_i xhat[b,i,m] U_in[i,r];  z *= S[r,m] (Hadamard: one launch with (r, m) as lanes);
            yhat[b,o,m] = sum_r z[b,r,m] U_out[o,r]."""
        kept, wsl = self._used_block(spatial)
        u_in, u_out = wsl.factors[0], wsl.factors[1]
        s = self._cp_mode_rows(wsl, kept)
        r, m = int(s.shape[0]), int(s.shape[1])
        ops = engine.EngineOps(self.fft_norm, self.engine_flags)
        xhat = ops.forward_transform(x, kept)
        b, ci = xhat.shape[:2]
        z = engine.mode_gemm(xhat.reshape(b, ci, m), u_in, m)                      # (B, R, M)
        z = engine.mode_gemm(z.reshape(b, 1, r * m), s.reshape(1, 1, r * m), r * m).reshape(b, r, m)
        yhat = engine.mode_gemm(z, u_out.transpose(0, 1), m)
        return ops.inverse_transform(yhat.reshape(b, u_out.shape[0], *kept), self.bias, spatial)

    # ---- tensor-train weights on the engine -----------------------------------------------------
    @staticmethod
    def _tt_dense(wsl, kept):
        """W[i, o, modes] of the used block from the TT cores, left to right, every step an sc_modegemm
        launch whose lanes are the (s_k, r_{k+1}) pairs of the core being absorbed -- the dense weight
        the reference's ``_contract_tt`` einsum (spectral_convolution.py:106-132) is equivalent to."""
        cores = list(wsl.factors)
        res = cores[0].reshape(cores[0].shape[1], cores[0].shape[2])              # (s_0, r_1)
        for g in cores[1:]:
            r, sk, rn = (int(v) for v in g.shape)
            lanes = sk * rn
            res = engine.mode_gemm(res, g.reshape(r, 1, lanes), lanes)            # (P, 1, s_k r_{k+1})
            res = res.reshape(-1, rn)
        return res.reshape(*[int(c.shape[1]) for c in cores])

    @staticmethod
    def _tucker_core_times_modes(wsl, kept):
        """T[f, g, modes] = core x_modes U_modes as a chain of sc_modegemm launches (autograd through
        ModeGemmFn).  rocBLAS' complex GEMMs behind torch.tensordot took 9-37 ms per call on these
        skinny shapes; the whole chain is ~0.3 ms here."""
        core = wsl.core
        mode_f = list(wsl.factors[2:])
        if len(mode_f) == 2 and core.dim() == 4:              # 2-D: core x_x U_x x_y U_y in one launch each way
            t3 = engine.tucker_modes_2d(core, mode_f[0], mode_f[1])
            if t3 is not None:
                return t3
        f, g = int(core.shape[0]), int(core.shape[1])
        ranks = [int(r) for r in core.shape[2:]]
        nd = len(mode_f)
        # last mode dim: lanes = its modes, the core is the mode-independent operand
        xk = core.reshape(-1, ranks[-1])
        u = mode_f[-1]                                       # (M_N, R_N)
        xk = engine.mode_gemm(xk, u.transpose(0, 1).unsqueeze(1), int(kept[-1]))     # (P, 1, M_N)
        tail = int(kept[-1])
        for d in range(nd - 2, -1, -1):                      # remaining mode dims: lanes = expanded tail
            lead = f * g
            for r in ranks[:d]:
                lead *= r
            xk = xk.reshape(lead, ranks[d], tail)
            xk = engine.mode_gemm(xk, mode_f[d].transpose(0, 1), tail)              # (lead, M_d, tail)
            tail *= int(kept[d])
        return xk.reshape(f, g, tail)

    def _tucker_dense(self, wsl, kept):
        """W[i, o, modes] of the used block from the factors, on the engine (implementation="reconstructed")."""
        t3 = self._tucker_core_times_modes(wsl, kept)
        m = int(t3.shape[2])
        w1 = engine.mode_gemm(wsl.factors[0], t3, m)                                  # (Cin, G, M)
        w = engine.mode_gemm(w1, wsl.factors[1].transpose(0, 1), m)                  # (Cin, Cout, M)
        return w.reshape(w.shape[0], w.shape[1], *kept)

    def _forward_tucker(self, x, spatial):
        """implementation="factorized" with a Tucker weight: the contraction never forms the dense
        weight.  Pairwise order of SURVEY.md section 8(a6) (the minimum-FLOP order of the reference's
        einsum 'abcd,fghi,bf,eg,ch,di->aecd', spectral_convolution.py:76-103):
            T[f,g,modes] = core x_modes U_modes     (batch independent)
            z[b,f,m] = sum_i xhat[b,i,m] U_in[i,f]; t[b,g,m] = sum_f z[b,f,m] T[f,g,m];
            yhat[b,o,m] = sum_g t[b,g,m] U_out[o,g]
        every step an sc_modegemm launch, autograd through sc_modegemm / sc_modegemm_msum (engine.TuckerChainFn)."""
        kept, wsl = self._used_block(spatial)
        u_in, u_out = wsl.factors[0], wsl.factors[1]
        t3 = self._tucker_core_times_modes(wsl, kept)
        m = int(t3.shape[2])
        ops = engine.EngineOps(self.fft_norm, self.engine_flags)
        xhat = ops.forward_transform(x, kept)               # (B, Cin, *kept) complex64
        b, ci = xhat.shape[:2]
        yhat = engine.tucker_chain(xhat.reshape(b, ci, m), u_in, t3, u_out)           # the three steps, one autograd node
        return ops.inverse_transform(yhat.reshape(b, u_out.shape[0], *kept), self.bias, spatial)
