"""ctypes binding of libsc_engine.so (C-ABI: include/sc_engine.h).

The product loads exactly one library: ``neuraloperator_amd/libsc_engine.so``, the hipcc
gfx950 build of ``csrc/sc_engine.cpp``.  If it is missing the import of the engine raises --
there is no CPU or PyTorch fallback for the hot path.
"""
import atexit
import ctypes
import os
import sys
from ctypes import (POINTER, Structure, byref, c_char_p, c_int, c_int32, c_int64, c_size_t,
                    c_void_p)

SC_MAX_DIMS = 4
SC_NORM = {"forward": 0, "backward": 1, "ortho": 2}
SC_FWD_SCALED, SC_FWD_ADJ_C2R = 0, 1
SC_INV_PADDED, SC_INV_ADJ_R2C = 0, 1
SC_PLAN_FORCE_GENERIC = 1
SC_PLAN_FFT_GEN2 = 2
SC_PLAN_NO_MDFT = 4
SC_PLAN_COMPLEX = 8
SC_PLAN_IO_BF16 = 16
SC_PLAN_NO_F2P_SMALL = 32
SC_PLAN_F2P_SMALL_ALWAYS = 64
SC_PLAN_NO_SPAN = 128
SC_PLAN_NO_MX_FFT = 256
SC_PLAN_MX_FFT_3TERM = 512
SC_FREQ_DROPPED = -(1 << 63)
SC_GEMM_FORCE_VALU = 1
SC_GEMM_STREAM_C = 2
SC_GEMM_PAIRED = 4
SC_GEMM_WIDE = 8
SC_GEMM_NO_STREAM = 16
SC_GEMM_F16 = 32
SC_GEMM_NO_SB = 64
SC_GEMM_SB_WM4 = 128
SC_GEMM_NO_FMX = 1 << 24
SC_GEMM_SB_ALT_ORDER = 1 << 25


def SC_GEMM_GRID(n):
    """flags bits 8..23 of sc_modegemm_desc: cap on the matrix-core kernel's workgroups."""
    return (int(n) & 0xffff) << 8

_HERE = os.path.dirname(os.path.abspath(__file__))
DEFAULT_LIB = os.path.join(_HERE, "libsc_engine.so")


class PlanDesc(Structure):
    _fields_ = [("ndim", c_int32), ("fft_norm", c_int32),
                ("spatial", c_int64 * SC_MAX_DIMS), ("kept", c_int64 * SC_MAX_DIMS),
                ("flags", c_int32), ("real_col", c_int32),
                ("freq", POINTER(c_int64) * SC_MAX_DIMS)]


class AdamwDesc(Structure):
    _fields_ = [("lr", ctypes.c_double), ("beta1", ctypes.c_double), ("beta2", ctypes.c_double),
                ("eps", ctypes.c_double), ("weight_decay", ctypes.c_double), ("step", c_int64),
                ("correct_bias", c_int32), ("reserved", c_int32)]


class SpectrumShards(Structure):           # sc_spectrum_shards
    _fields_ = [("n_blocks", c_int64), ("rows", c_int64), ("block_stride", c_int64)]


class ModeGemmDesc(Structure):
    _fields_ = [("P", c_int64), ("Q", c_int64), ("R", c_int64), ("n_modes", c_int64),
                ("a_sp", c_int64), ("a_sr", c_int64), ("a_sm", c_int64),
                ("b_sr", c_int64), ("b_sq", c_int64), ("b_sm", c_int64),
                ("c_sp", c_int64), ("c_sq", c_int64), ("c_sm", c_int64),
                ("conj_a", c_int32), ("conj_b", c_int32),
                ("accumulate", c_int32), ("flags", c_int32),
                ("b_idx", c_void_p), ("c_idx", c_void_p),
                ("a_sg", c_int64), ("b_sg", c_int64), ("c_sg", c_int64)]


class LayerDesc(Structure):
    _fields_ = [("batch", c_int32), ("cin", c_int32), ("cout", c_int32), ("reserved", c_int32),
                ("w_extent", c_int64 * SC_MAX_DIMS), ("w_start", c_int64 * SC_MAX_DIMS)]


class Epilogue(Structure):
    _fields_ = [("skip", c_void_p), ("preact", c_void_p), ("act", c_int32), ("reserved", c_int32)]


SC_PLX_XACT, SC_PLX_ACT, SC_PLX_PRO, SC_PLX_XGRAD = 1, 2, 4, 8     # sc_plinx_desc.flags
SC_ACT_NONE, SC_ACT_GELU, SC_ACT_GELU_DGRAD = 0, 1, 2


class TuckerDesc(Structure):
    _fields_ = [("fg", c_int64), ("rx", c_int64), ("ry", c_int64), ("mx", c_int64), ("my", c_int64)]


class TuckerChainDesc(Structure):
    _fields_ = [("batch", c_int64), ("c_in", c_int64), ("c_out", c_int64), ("r_in", c_int64), ("r_out", c_int64),
                ("n_modes", c_int64)]


class PeerExchangeDesc(Structure):
    _fields_ = [("world", c_int32), ("rank", c_int32), ("block_bytes", c_int64), ("peer_window", c_void_p * 8)]


class PlinDesc(Structure):
    _fields_ = [("batch", c_int64), ("c_in", c_int64), ("c_out", c_int64), ("spatial", c_int64)]


class PlinxDesc(Structure):                                 # sc_plinx_desc (round 6)
    _fields_ = [("batch", c_int64), ("c_in", c_int64), ("c_out", c_int64), ("spatial", c_int64), ("flags", c_int32)]


class PmlpDesc(Structure):
    _fields_ = [("batch", c_int64), ("c_in", c_int64), ("c_hid", c_int64), ("c_out", c_int64), ("spatial", c_int64),
                ("act", c_int32), ("reserved", c_int32)]


class EngineError(RuntimeError):
    pass


_SHUTDOWN = False          # set at interpreter exit: device memory goes with the process, the HIP runtime may be gone


@atexit.register
def _mark_shutdown():
    global _SHUTDOWN
    _SHUTDOWN = True


class PlanHandle:
    """Owner of one sc_plan*.  ctypes passes ``_as_parameter_``; the plan (its device twiddle / index tables) is
    released by plan_destroy() or when the last reference goes away -- the plan cache of engine.py only drops ITS
    reference on eviction, autograd contexts that still hold the plan keep it alive."""

    def __init__(self, lib, ptr):
        self._lib, self._as_parameter_ = lib, ptr

    def destroy(self, _finalizing=sys.is_finalizing):
        # (the default argument keeps sys.is_finalizing reachable while module globals are being torn down: a
        # handle that outlives its module -- held by a script's globals -- is finalised after `sys` became None)
        ptr, self._as_parameter_ = self._as_parameter_, None
        try:
            if ptr is not None and self._lib is not None and not _SHUTDOWN and not _finalizing():
                self._lib.sc_plan_destroy(ptr)
        except Exception:                # interpreter shutdown
            pass

    def __del__(self):
        self.destroy()


class ScEngineLib:
    """Thin typed wrapper over the shared library.  Pointers are raw integers
    (``tensor.data_ptr()``); ``stream`` is a raw hipStream_t (0 = null stream)."""

    # every symbol include/sc_engine.h declares
    SYMBOLS = ["sc_plan_create", "sc_plan_destroy", "sc_plan_workspace_bytes", "sc_plan_is_fast",
               "sc_transform_forward", "sc_transform_inverse", "sc_modegemm",
               "sc_modegemm_msum", "sc_modegemm_msum_ws", "sc_modegemm_msum_workspace_bytes", "sc_modegemm_msum_path", "sc_modegemm_uses_matrix_cores", "sc_modegemm_path", "sc_bias_grad", "sc_adamw_step",
               "sc_layer_workspace_bytes", "sc_layer_forward", "sc_layer_backward",
               "sc_last_error", "sc_version", "sc_plan_kernel_name", "sc_transform_inverse_ex",
               "sc_layer_forward_ex", "sc_round_f16", "sc_pointwise_mlp_forward", "sc_pointwise_block_forward",
               "sc_pointwise_mlp_backward", "sc_pointwise_mlp_workspace_bytes", "sc_pointwise_linear_forward",
               "sc_pointwise_linear_backward", "sc_pointwise_linear_workspace_bytes", "sc_layer_backward_ex",
               "sc_pointwise_mlp_backward_ex", "sc_tucker_modes_supported", "sc_tucker_modes_forward",
               "sc_tucker_modes_backward", "sc_tucker_modes_workspace_bytes", "sc_modegemm_pair",
               "sc_modegemm_pair_fused", "sc_modegemm_pair_path", "sc_plan_workspace_bytes_sharded", "sc_transform_forward_sharded",
               "sc_transform_inverse_sharded", "sc_bias_grad_sharded", "sc_tucker_chain_forward",
               "sc_tucker_chain_backward", "sc_tucker_chain_workspace_bytes", "sc_tucker_chain_fused_supported",
               "sc_tucker_chain_t3m_bytes", "sc_tucker_chain_forward_fused", "sc_tucker_chain_backward_fused",
               "sc_tucker_chain_backward_fused_workspace_bytes", "sc_peer_window_alloc", "sc_peer_window_open",
               "sc_peer_window_close", "sc_peer_window_free", "sc_peer_all_to_all", "sc_peer_window_control", "sc_pointwise_linear_forward_ex",
               "sc_pointwise_linear_workspace_bytes_ex", "sc_pointwise_linear_backward_ex", "sc_pointwise_block_backward",
               "sc_pointwise_block_backward_supported"]

    def __init__(self, path=DEFAULT_LIB):
        if not os.path.isfile(path):
            raise EngineError(
                f"{path} not found: the HIP engine is not built. Run "
                "`python -c 'import __graft_entry__ as g; g.build()'` (needs hipcc). "
                "There is no fallback path.")
        self.path = path
        self.lib = ctypes.CDLL(path)
        L = self.lib
        for s in self.SYMBOLS:
            if not hasattr(L, s):
                raise EngineError(f"{path} does not export {s}")
        L.sc_plan_create.argtypes = [POINTER(c_void_p), POINTER(PlanDesc)]
        L.sc_plan_create.restype = c_int
        L.sc_plan_destroy.argtypes = [c_void_p]
        L.sc_plan_destroy.restype = None
        L.sc_plan_workspace_bytes.argtypes = [c_void_p, c_int64]
        L.sc_plan_workspace_bytes.restype = c_size_t
        L.sc_plan_is_fast.argtypes = [c_void_p]
        L.sc_plan_is_fast.restype = c_int
        L.sc_transform_forward.argtypes = [c_void_p, c_int, c_void_p, c_void_p, c_int64,
                                           c_void_p, c_void_p]
        L.sc_transform_forward.restype = c_int
        L.sc_transform_inverse.argtypes = [c_void_p, c_int, c_void_p, c_void_p, c_int64,
                                           c_void_p, c_int64, c_void_p, c_void_p]
        L.sc_transform_inverse.restype = c_int
        L.sc_modegemm.argtypes = [POINTER(ModeGemmDesc), c_void_p, c_void_p, c_void_p, c_void_p]
        L.sc_modegemm.restype = c_int
        L.sc_modegemm_pair.argtypes = [POINTER(ModeGemmDesc), c_void_p, c_void_p, c_void_p,
                                       POINTER(ModeGemmDesc), c_void_p, c_void_p, c_void_p, c_void_p]
        L.sc_modegemm_pair.restype = c_int
        L.sc_modegemm_pair_fused.argtypes = [POINTER(ModeGemmDesc), POINTER(ModeGemmDesc)]
        L.sc_modegemm_pair_fused.restype = c_int
        L.sc_modegemm_pair_path.argtypes = [POINTER(ModeGemmDesc), POINTER(ModeGemmDesc)]
        L.sc_modegemm_pair_path.restype = c_int
        L.sc_modegemm_msum.argtypes = [POINTER(ModeGemmDesc), c_void_p, c_void_p, c_void_p, c_void_p]
        L.sc_modegemm_msum.restype = c_int
        L.sc_modegemm_msum_workspace_bytes.argtypes = [POINTER(ModeGemmDesc)]
        L.sc_modegemm_msum_workspace_bytes.restype = c_size_t
        L.sc_modegemm_msum_ws.argtypes = [POINTER(ModeGemmDesc), c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]
        L.sc_modegemm_msum_ws.restype = c_int
        L.sc_modegemm_msum_path.argtypes = [POINTER(ModeGemmDesc)]
        L.sc_modegemm_msum_path.restype = c_int
        L.sc_modegemm_uses_matrix_cores.argtypes = [POINTER(ModeGemmDesc)]
        L.sc_modegemm_uses_matrix_cores.restype = c_int
        L.sc_modegemm_path.argtypes = [POINTER(ModeGemmDesc)]
        L.sc_modegemm_path.restype = c_int
        L.sc_bias_grad.argtypes = [c_void_p, c_void_p, c_int64, c_int64, c_void_p, c_void_p]
        L.sc_bias_grad.restype = c_int
        L.sc_plan_workspace_bytes_sharded.argtypes = [c_void_p, c_int64]
        L.sc_plan_workspace_bytes_sharded.restype = c_size_t
        L.sc_transform_forward_sharded.argtypes = [c_void_p, c_int, c_void_p, c_void_p, c_int64,
                                                   POINTER(SpectrumShards), c_void_p, c_void_p]
        L.sc_transform_forward_sharded.restype = c_int
        L.sc_transform_inverse_sharded.argtypes = [c_void_p, c_int, c_void_p, c_void_p, c_int64, c_void_p, c_int64,
                                                   POINTER(SpectrumShards), c_void_p, c_void_p]
        L.sc_transform_inverse_sharded.restype = c_int
        L.sc_bias_grad_sharded.argtypes = [c_void_p, c_void_p, c_int64, c_int64, POINTER(SpectrumShards), c_void_p,
                                           c_void_p]
        L.sc_bias_grad_sharded.restype = c_int
        L.sc_adamw_step.argtypes = [POINTER(AdamwDesc), c_void_p, c_void_p, c_void_p, c_void_p, c_int64,
                                    c_int, c_void_p]
        L.sc_adamw_step.restype = c_int
        L.sc_layer_workspace_bytes.argtypes = [c_void_p, POINTER(LayerDesc)]
        L.sc_layer_workspace_bytes.restype = c_size_t
        L.sc_layer_forward.argtypes = [c_void_p, POINTER(LayerDesc)] + [c_void_p] * 7
        L.sc_layer_forward.restype = c_int
        L.sc_transform_inverse_ex.argtypes = [c_void_p, c_int, c_void_p, c_void_p, c_int64, POINTER(Epilogue),
                                              c_void_p, c_int64, c_void_p, c_void_p]
        L.sc_transform_inverse_ex.restype = c_int
        L.sc_layer_forward_ex.argtypes = [c_void_p, POINTER(LayerDesc), c_void_p, c_void_p, c_void_p,
                                          POINTER(Epilogue), c_void_p, c_void_p, c_void_p, c_void_p]
        L.sc_layer_forward_ex.restype = c_int
        L.sc_layer_backward.argtypes = [c_void_p, POINTER(LayerDesc)] + [c_void_p] * 8
        L.sc_layer_backward_ex.argtypes = [c_void_p, POINTER(LayerDesc)] + [c_void_p] * 9
        L.sc_layer_backward_ex.restype = c_int
        L.sc_layer_backward.restype = c_int
        L.sc_pointwise_mlp_forward.argtypes = [POINTER(PmlpDesc)] + [c_void_p] * 9
        L.sc_pointwise_mlp_forward.restype = c_int
        L.sc_pointwise_block_forward.argtypes = [POINTER(PmlpDesc)] + [c_void_p] * 13
        L.sc_pointwise_block_forward.restype = c_int
        L.sc_pointwise_mlp_workspace_bytes.argtypes = [POINTER(PmlpDesc)]
        L.sc_pointwise_mlp_workspace_bytes.restype = c_size_t
        L.sc_pointwise_mlp_backward.argtypes = [POINTER(PmlpDesc)] + [c_void_p] * 17
        L.sc_pointwise_mlp_backward.restype = c_int
        L.sc_pointwise_mlp_backward_ex.argtypes = [POINTER(PmlpDesc)] + [c_void_p] * 18
        L.sc_pointwise_mlp_backward_ex.restype = c_int
        L.sc_pointwise_linear_forward.argtypes = [POINTER(PlinDesc)] + [c_void_p] * 5
        L.sc_pointwise_linear_forward.restype = c_int
        L.sc_pointwise_linear_workspace_bytes.argtypes = [POINTER(PlinDesc)]
        L.sc_pointwise_linear_workspace_bytes.restype = c_size_t
        L.sc_pointwise_block_backward_supported.argtypes = [POINTER(PmlpDesc)]
        L.sc_pointwise_block_backward_supported.restype = c_int
        L.sc_pointwise_block_backward.argtypes = [POINTER(PmlpDesc)] + [c_void_p] * 19
        L.sc_pointwise_block_backward.restype = c_int
        L.sc_pointwise_linear_forward_ex.argtypes = [POINTER(PlinxDesc)] + [c_void_p] * 8
        L.sc_pointwise_linear_forward_ex.restype = c_int
        L.sc_pointwise_linear_workspace_bytes_ex.argtypes = [POINTER(PlinxDesc)]
        L.sc_pointwise_linear_workspace_bytes_ex.restype = c_size_t
        L.sc_pointwise_linear_backward_ex.argtypes = [POINTER(PlinxDesc)] + [c_void_p] * 15
        L.sc_pointwise_linear_backward_ex.restype = c_int
        L.sc_pointwise_linear_backward.argtypes = [POINTER(PlinDesc)] + [c_void_p] * 9
        L.sc_pointwise_linear_backward.restype = c_int
        L.sc_tucker_modes_supported.argtypes = [POINTER(TuckerDesc)]
        L.sc_tucker_modes_supported.restype = c_int
        L.sc_tucker_modes_forward.argtypes = [POINTER(TuckerDesc)] + [c_void_p] * 5
        L.sc_tucker_modes_forward.restype = c_int
        L.sc_tucker_modes_workspace_bytes.argtypes = [POINTER(TuckerDesc)]
        L.sc_tucker_modes_workspace_bytes.restype = c_size_t
        L.sc_tucker_modes_backward.argtypes = [POINTER(TuckerDesc)] + [c_void_p] * 9
        L.sc_tucker_modes_backward.restype = c_int
        L.sc_tucker_chain_forward.argtypes = [POINTER(TuckerChainDesc)] + [c_void_p] * 8
        L.sc_tucker_chain_forward.restype = c_int
        L.sc_tucker_chain_workspace_bytes.argtypes = [POINTER(TuckerChainDesc)]
        L.sc_tucker_chain_workspace_bytes.restype = c_size_t
        L.sc_tucker_chain_backward.argtypes = [POINTER(TuckerChainDesc)] + [c_void_p] * 12 + [c_size_t, c_void_p]
        L.sc_tucker_chain_backward.restype = c_int
        L.sc_peer_window_alloc.argtypes = [c_size_t, POINTER(c_void_p), c_void_p]
        L.sc_peer_window_alloc.restype = c_int
        L.sc_peer_window_open.argtypes = [c_void_p, POINTER(c_void_p)]
        L.sc_peer_window_open.restype = c_int
        L.sc_peer_window_close.argtypes = [c_void_p]
        L.sc_peer_window_close.restype = c_int
        L.sc_peer_window_free.argtypes = [c_void_p]
        L.sc_peer_window_free.restype = c_int
        L.sc_peer_all_to_all.argtypes = [POINTER(PeerExchangeDesc), c_void_p, c_void_p, c_void_p]
        L.sc_peer_all_to_all.restype = c_int
        L.sc_peer_window_control.argtypes = [c_void_p, c_int64, POINTER(ctypes.c_int32)]
        L.sc_peer_window_control.restype = c_int
        L.sc_tucker_chain_fused_supported.argtypes = [POINTER(TuckerChainDesc)]
        L.sc_tucker_chain_fused_supported.restype = c_int
        L.sc_tucker_chain_t3m_bytes.argtypes = [POINTER(TuckerChainDesc)]
        L.sc_tucker_chain_t3m_bytes.restype = c_size_t
        L.sc_tucker_chain_forward_fused.argtypes = [POINTER(TuckerChainDesc)] + [c_void_p] * 9
        L.sc_tucker_chain_forward_fused.restype = c_int
        L.sc_tucker_chain_backward_fused_workspace_bytes.argtypes = [POINTER(TuckerChainDesc)]
        L.sc_tucker_chain_backward_fused_workspace_bytes.restype = c_size_t
        L.sc_tucker_chain_backward_fused.argtypes = [POINTER(TuckerChainDesc)] + [c_void_p] * 12 + [c_size_t, c_void_p]
        L.sc_tucker_chain_backward_fused.restype = c_int
        L.sc_round_f16.argtypes = [c_void_p, c_void_p, c_int64, c_void_p]
        L.sc_round_f16.restype = c_int
        L.sc_last_error.restype = c_char_p
        L.sc_version.restype = c_char_p
        L.sc_plan_kernel_name.argtypes = [c_void_p, c_int]
        L.sc_plan_kernel_name.restype = c_char_p

    # -- helpers -----------------------------------------------------------------------
    def _check(self, rc):
        if rc != 0:
            raise EngineError(self.lib.sc_last_error().decode())

    def version(self):
        return self.lib.sc_version().decode()

    # -- plan ---------------------------------------------------------------------------
    def plan_create(self, spatial, kept, fft_norm="forward", flags=0, freq=None, real_col=0):
        """freq: optional per-dim frequency maps (None = the default same-grid rule, an entry of None
        or SC_FREQ_DROPPED = row falls off the grid); see include/sc_engine.h and modes.py."""
        d = PlanDesc()
        d.ndim = len(spatial)
        if not 1 <= d.ndim <= SC_MAX_DIMS:
            raise EngineError(f"spatial rank {d.ndim} unsupported (1..{SC_MAX_DIMS})")
        d.fft_norm = SC_NORM[fft_norm]
        for i, (n, k) in enumerate(zip(spatial, kept)):
            d.spatial[i] = int(n)
            d.kept[i] = int(k)
        d.flags = flags
        d.real_col = int(real_col)
        keep = []                                            # host arrays must outlive the call only
        if freq is not None:
            for i, f in enumerate(freq):
                if f is None:
                    continue
                if len(f) != int(kept[i]):
                    raise EngineError(f"frequency map of dim {i} has {len(f)} entries, kept = {kept[i]}")
                arr = (c_int64 * len(f))(*[SC_FREQ_DROPPED if v is None else int(v) for v in f])
                keep.append(arr)
                d.freq[i] = ctypes.cast(arr, POINTER(c_int64))
        h = c_void_p()
        self._check(self.lib.sc_plan_create(byref(h), byref(d)))
        return PlanHandle(self.lib, h)

    def plan_destroy(self, plan):
        plan.destroy()

    def plan_workspace_bytes(self, plan, n_images):
        return int(self.lib.sc_plan_workspace_bytes(plan, n_images))

    def plan_is_fast(self, plan):
        return bool(self.lib.sc_plan_is_fast(plan))

    def plan_kernel_name(self, plan, which):
        return self.lib.sc_plan_kernel_name(plan, which).decode()

    # -- sharded spectra (mode-parallel layers): the all-to-all buffer [n_blocks][n_images][rows][rest] in place
    def plan_workspace_bytes_sharded(self, plan, n_images):
        return int(self.lib.sc_plan_workspace_bytes_sharded(plan, n_images))

    @staticmethod
    def shards(n_blocks, rows, block_stride):
        return SpectrumShards(int(n_blocks), int(rows), int(block_stride))

    def transform_forward_sharded(self, plan, mode, x_ptr, xhat_ptr, n_images, shards, ws_ptr, stream=0):
        self._check(self.lib.sc_transform_forward_sharded(plan, mode, x_ptr, xhat_ptr, n_images, byref(shards),
                                                          ws_ptr, stream))

    def transform_inverse_sharded(self, plan, mode, yhat_ptr, bias_ptr, channels, y_ptr, n_images, shards, ws_ptr,
                                  stream=0):
        self._check(self.lib.sc_transform_inverse_sharded(plan, mode, yhat_ptr, bias_ptr, channels, y_ptr, n_images,
                                                          byref(shards), ws_ptr, stream))

    def bias_grad_sharded(self, plan, ghat_ptr, batch, channels, shards, gbias_ptr, stream=0):
        self._check(self.lib.sc_bias_grad_sharded(plan, ghat_ptr, batch, channels, byref(shards), gbias_ptr, stream))

    # -- stages ------------------------------------------------------------------------
    def transform_forward(self, plan, mode, x_ptr, xhat_ptr, n_images, ws_ptr, stream=0):
        self._check(self.lib.sc_transform_forward(plan, mode, x_ptr, xhat_ptr, n_images,
                                                  ws_ptr, stream))

    def transform_inverse(self, plan, mode, yhat_ptr, bias_ptr, channels, y_ptr, n_images,
                          ws_ptr, stream=0):
        self._check(self.lib.sc_transform_inverse(plan, mode, yhat_ptr, bias_ptr, channels,
                                                  y_ptr, n_images, ws_ptr, stream))

    def modegemm(self, a_ptr, b_ptr, c_ptr, stream=0, **kw):
        self._check(self.lib.sc_modegemm(byref(self._gemm_desc(kw)), a_ptr, b_ptr, c_ptr, stream))

    _DESC_CACHE = {}

    @staticmethod
    def _gemm_desc(kw):
        """The descriptor of a contraction, cached by value: building a 22-field ctypes structure through setattr costs
        about as much host time as the launch itself (a factorized layer step issues ~20 of them)."""
        key = tuple(kw.items())
        d = ScEngineLib._DESC_CACHE.get(key)
        if d is None:
            if len(ScEngineLib._DESC_CACHE) > 4096:
                ScEngineLib._DESC_CACHE.clear()
            d = ModeGemmDesc()
            for k, v in kw.items():
                setattr(d, k, v)
            ScEngineLib._DESC_CACHE[key] = d
        return d

    def modegemm_pair(self, kw0, a0, b0, c0, kw1, a1, b1, c1, stream=0):
        """sc_modegemm(kw0 ...) and sc_modegemm(kw1 ...), one launch when the pair qualifies (a layer's backward)."""
        d0, d1 = self._gemm_desc(kw0), self._gemm_desc(kw1)
        self._check(self.lib.sc_modegemm_pair(byref(d0), a0, b0, c0, byref(d1), a1, b1, c1, stream))

    def modegemm_pair_fused(self, kw0, kw1):
        d0, d1 = self._gemm_desc(kw0), self._gemm_desc(kw1)
        return bool(self.lib.sc_modegemm_pair_fused(byref(d0), byref(d1)))

    def modegemm_pair_path(self, kw0, kw1):
        """2 = one pass over the weight (k_modegemm_sb_bwd), 1 = one k_modegemm_dma_bwd launch, 0 = two launches."""
        d0, d1 = self._gemm_desc(kw0), self._gemm_desc(kw1)
        return int(self.lib.sc_modegemm_pair_path(byref(d0), byref(d1)))

    def pointwise_mlp_forward(self, batch, c_in, c_hid, c_out, spatial, act, x, w1, b1, w2, b2, skip, gate, out, stream=0):
        d = PmlpDesc(batch, c_in, c_hid, c_out, spatial, act, 0)
        self._check(self.lib.sc_pointwise_mlp_forward(byref(d), x, w1, b1, w2, b2, skip, gate, out, stream))

    def pointwise_mlp_workspace_bytes(self, batch, c_in, c_hid, c_out, spatial, act):
        d = PmlpDesc(batch, c_in, c_hid, c_out, spatial, act, 0)
        return int(self.lib.sc_pointwise_mlp_workspace_bytes(byref(d)))

    def pointwise_block_forward(self, batch, c, c_hid, spatial, act, conv, x, ws, bs, w1, b1, w2, b2, gate, y, pre, out,
                                stream=0):
        """s = conv + (ws x + bs); y = act(s); out = act(W2 gelu(W1 y + b1) + b2 + gate x) in one pass (y, pre stored)."""
        d = PmlpDesc(batch, c, c_hid, c, spatial, act, 0)
        self._check(self.lib.sc_pointwise_block_forward(byref(d), conv, x, ws, bs, w1, b1, w2, b2, gate, y, pre, out, stream))

    def pointwise_mlp_backward(self, batch, c_in, c_hid, c_out, spatial, act, x, w1, b1, w2, b2, skip, gate, gout,
                               gx, gw1, gb1, gw2, gb2, gskip, ggate, ws, stream=0, x_pre=0):
        d = PmlpDesc(batch, c_in, c_hid, c_out, spatial, act, 0)
        self._check(self.lib.sc_pointwise_mlp_backward_ex(byref(d), x, x_pre, w1, b1, w2, b2, skip, gate, gout, gx, gw1,
                                                          gb1, gw2, gb2, gskip, ggate, ws, stream))

    def pointwise_block_backward_supported(self, batch, c, c_hid, spatial):
        return bool(self.lib.sc_pointwise_block_backward_supported(byref(PmlpDesc(batch, c, c_hid, c, spatial, 0, 0))))

    def pointwise_block_backward(self, batch, c, c_hid, spatial, act, y, y_pre, x, ws_lin, w1, b1, w2, b2, gate, gout, gz, gin,
                                 gw1, gb1, gw2, gb2, ggate, ws, stream=0):
        """the MLP pass of a block's backward + the data path of its linear skip (include/sc_engine.h, round 6)"""
        d = PmlpDesc(batch, c, c_hid, c, spatial, act, 0)
        self._check(self.lib.sc_pointwise_block_backward(byref(d), y, y_pre, x, ws_lin, w1, b1, w2, b2, gate, gout, gz, gin,
                                                         gw1, gb1, gw2, gb2, ggate, ws, stream))

    def pointwise_linear_forward(self, batch, c_in, c_out, spatial, x, w, bias, out, stream=0):
        d = PlinDesc(batch, c_in, c_out, spatial)
        self._check(self.lib.sc_pointwise_linear_forward(byref(d), x, w, bias, out, stream))

    def pointwise_linear_workspace_bytes(self, batch, c_in, c_out, spatial):
        d = PlinDesc(batch, c_in, c_out, spatial)
        return int(self.lib.sc_pointwise_linear_workspace_bytes(byref(d)))

    def pointwise_linear_backward(self, batch, c_in, c_out, spatial, x, w, gout, gx, gw, gbias, ws, stream=0, addend=0):
        d = PlinDesc(batch, c_in, c_out, spatial)
        self._check(self.lib.sc_pointwise_linear_backward(byref(d), x, w, gout, addend, gx, gw, gbias, ws, stream))

    # ---- 1 x 1 maps with the block's pointwise operations in their load / store paths (round 6; include/sc_engine.h)
    def pointwise_linear_forward_ex(self, batch, c_in, c_out, spatial, flags, x, w, bias, skip, gate, out, pre_out=0, stream=0):
        d = PlinxDesc(batch, c_in, c_out, spatial, flags)
        self._check(self.lib.sc_pointwise_linear_forward_ex(byref(d), x, w, bias, skip, gate, out, pre_out, stream))

    def pointwise_linear_workspace_bytes_ex(self, batch, c_in, c_out, spatial):
        return int(self.lib.sc_pointwise_linear_workspace_bytes_ex(byref(PlinxDesc(batch, c_in, c_out, spatial, 0))))

    def pointwise_linear_backward_ex(self, batch, c_in, c_out, spatial, flags, x, w, gout, pre, xg, skip, gate, addend, gx, gw,
                                     gbias, gskip, ggate, ws, stream=0):
        d = PlinxDesc(batch, c_in, c_out, spatial, flags)
        self._check(self.lib.sc_pointwise_linear_backward_ex(byref(d), x, w, gout, pre, xg, skip, gate, addend, gx, gw, gbias,
                                                             gskip, ggate, ws, stream))

    def tucker_modes_supported(self, fg, rx, ry, mx, my):
        return bool(self.lib.sc_tucker_modes_supported(byref(TuckerDesc(fg, rx, ry, mx, my))))

    def tucker_modes_forward(self, fg, rx, ry, mx, my, core, ux, uy, t, stream=0):
        self._check(self.lib.sc_tucker_modes_forward(byref(TuckerDesc(fg, rx, ry, mx, my)), core, ux, uy, t, stream))

    def tucker_modes_workspace_bytes(self, fg, rx, ry, mx, my):
        return int(self.lib.sc_tucker_modes_workspace_bytes(byref(TuckerDesc(fg, rx, ry, mx, my))))

    def tucker_modes_backward(self, fg, rx, ry, mx, my, core, ux, uy, gt, gcore, gux, guy, ws, stream=0):
        self._check(self.lib.sc_tucker_modes_backward(byref(TuckerDesc(fg, rx, ry, mx, my)), core, ux, uy, gt, gcore, gux,
                                                      guy, ws, stream))

    def tucker_chain_forward(self, dims, xhat, u_in, t3, u_out, z, t, yhat, stream=0):
        """dims = (batch, c_in, c_out, r_in, r_out, n_modes); device pointers of contiguous complex64 arrays"""
        self._check(self.lib.sc_tucker_chain_forward(byref(TuckerChainDesc(*dims)), xhat, u_in, t3, u_out, z, t, yhat, stream))

    def tucker_chain_workspace_bytes(self, dims):
        return int(self.lib.sc_tucker_chain_workspace_bytes(byref(TuckerChainDesc(*dims))))

    def tucker_chain_backward(self, dims, xhat, u_in, t3, u_out, z, t, gy, gxhat, gu_in, gt3, gu_out, ws, ws_bytes, stream=0):
        """null (0) gradient pointers skip that gradient"""
        self._check(self.lib.sc_tucker_chain_backward(byref(TuckerChainDesc(*dims)), xhat, u_in, t3, u_out, z, t, gy, gxhat,
                                                      gu_in, gt3, gu_out, ws, ws_bytes, stream))

    # ---- peer-store exchange (round 5; include/sc_engine.h)
    def peer_window_alloc(self, data_bytes):
        """-> (device pointer, 64-byte IPC handle) of a fresh fine-grained window"""
        import ctypes
        ptr, handle = c_void_p(), ctypes.create_string_buffer(64)
        self._check(self.lib.sc_peer_window_alloc(data_bytes, byref(ptr), handle))
        return int(ptr.value), handle.raw

    def peer_window_open(self, handle):
        import ctypes
        ptr, buf = c_void_p(), ctypes.create_string_buffer(bytes(handle), 64)
        self._check(self.lib.sc_peer_window_open(buf, byref(ptr)))
        return int(ptr.value)

    def peer_window_close(self, ptr):
        self._check(self.lib.sc_peer_window_close(ptr))

    def peer_window_free(self, ptr):
        self._check(self.lib.sc_peer_window_free(ptr))

    def peer_window_control(self, own_window, spin_budget_ms=-1):
        """set the spin budget of the waits on this rank's window (ms; 0 = unbounded, < 0 = unchanged); -> the error word a
        timed-out wait left (0 = none, 1 + p = peer p's flag never came), cleared by the read"""
        import ctypes
        err = ctypes.c_int32(0)
        self._check(self.lib.sc_peer_window_control(own_window, spin_budget_ms, byref(err)))
        return int(err.value)

    def peer_all_to_all(self, world, rank, block_bytes, peer_windows, send, recv, stream=0):
        d = PeerExchangeDesc(world, rank, block_bytes, (c_void_p * 8)(*(list(peer_windows) + [None] * (8 - len(peer_windows)))))
        self._check(self.lib.sc_peer_all_to_all(byref(d), send, recv, stream))

    # ---- the chain as one launch each way (round 5; include/sc_engine.h)
    def tucker_chain_fused_supported(self, dims):
        return bool(self.lib.sc_tucker_chain_fused_supported(byref(TuckerChainDesc(*dims))))

    def tucker_chain_t3m_bytes(self, dims):
        return int(self.lib.sc_tucker_chain_t3m_bytes(byref(TuckerChainDesc(*dims))))

    def tucker_chain_forward_fused(self, dims, xhat, u_in, t3, u_out, t3m, z, t, yhat, stream=0):
        """t3m (n_modes, r_in, r_out): written here, read by tucker_chain_backward_fused"""
        self._check(self.lib.sc_tucker_chain_forward_fused(byref(TuckerChainDesc(*dims)), xhat, u_in, t3, u_out, t3m, z, t,
                                                           yhat, stream))

    def tucker_chain_backward_fused_workspace_bytes(self, dims):
        return int(self.lib.sc_tucker_chain_backward_fused_workspace_bytes(byref(TuckerChainDesc(*dims))))

    def tucker_chain_backward_fused(self, dims, xhat, u_in, t3m, u_out, z, t, gy, gxhat, gu_in, gt3, gu_out, ws, ws_bytes,
                                    stream=0):
        """gt3 is required; null (0) gxhat / gu_in / gu_out skip that gradient"""
        self._check(self.lib.sc_tucker_chain_backward_fused(byref(TuckerChainDesc(*dims)), xhat, u_in, t3m, u_out, z, t, gy,
                                                            gxhat, gu_in, gt3, gu_out, ws, ws_bytes, stream))

    def round_f16(self, in_ptr, out_ptr, n, stream=0):
        """out = float16(in) in fp32 storage (the cast points of fno_block_precision half / mixed)."""
        self._check(self.lib.sc_round_f16(in_ptr, out_ptr, n, stream))

    def modegemm_msum(self, a_ptr, b_ptr, c_ptr, stream=0, **kw):
        self._check(self.lib.sc_modegemm_msum(byref(self._gemm_desc(kw)), a_ptr, b_ptr, c_ptr, stream))

    def modegemm_msum_workspace_bytes(self, **kw):
        """Bytes of workspace sc_modegemm_msum_ws needs for this problem (0: empty problem)."""
        return int(self.lib.sc_modegemm_msum_workspace_bytes(byref(self._gemm_desc(kw))))

    def modegemm_msum_path(self, **kw):
        """1: sc_modegemm_msum_ws runs the matrix-core kernel, 0: the slot form of the VALU kernel."""
        return int(self.lib.sc_modegemm_msum_path(byref(self._gemm_desc(kw))))

    def modegemm_msum_ws(self, a_ptr, b_ptr, c_ptr, ws_ptr, ws_bytes, stream=0, **kw):
        """C[p, q] = sum over modes and r, OVERWRITTEN (matrix-core kernel + fixed-order reduction)."""
        self._check(self.lib.sc_modegemm_msum_ws(byref(self._gemm_desc(kw)), a_ptr, b_ptr, c_ptr, ws_ptr, ws_bytes, stream))

    def modegemm_uses_matrix_cores(self, **kw):
        d = ModeGemmDesc()
        for k, v in kw.items():
            setattr(d, k, v)
        return bool(self.lib.sc_modegemm_uses_matrix_cores(byref(d)))

    def modegemm_path(self, **kw):
        """0 VALU kernel, 1 k_modegemm_mfma, 2 k_modegemm_s8 (for 16-byte aligned operands)."""
        d = ModeGemmDesc()
        for k, v in kw.items():
            setattr(d, k, v)
        return int(self.lib.sc_modegemm_path(byref(d)))

    def adamw_step(self, p_ptr, g_ptr, m_ptr, v_ptr, n, is_complex, stream=0, *, lr, beta1, beta2, eps,
                   weight_decay, correct_bias, step):
        d = AdamwDesc(lr, beta1, beta2, eps, weight_decay, int(step), int(bool(correct_bias)), 0)
        self._check(self.lib.sc_adamw_step(byref(d), p_ptr, g_ptr, m_ptr, v_ptr, n, int(bool(is_complex)), stream))

    def bias_grad(self, plan, ghat_ptr, batch, channels, gbias_ptr, stream=0):
        self._check(self.lib.sc_bias_grad(plan, ghat_ptr, batch, channels, gbias_ptr, stream))

    # -- fused dense layer -------------------------------------------------------------
    @staticmethod
    def layer_desc(batch, cin, cout, w_extent, w_start):
        L = LayerDesc()
        L.batch, L.cin, L.cout = batch, cin, cout
        for i, (e, s) in enumerate(zip(w_extent, w_start)):
            L.w_extent[i] = int(e)
            L.w_start[i] = int(s)
        return L

    def layer_workspace_bytes(self, plan, L):
        return int(self.lib.sc_layer_workspace_bytes(plan, byref(L)))

    def layer_forward(self, plan, L, x, w, bias, y, xhat_saved, ws, stream=0):
        self._check(self.lib.sc_layer_forward(plan, byref(L), x, w, bias, y, xhat_saved, ws,
                                              stream))

    def layer_forward_ex(self, plan, L, x, w, bias, skip, preact, act, y, xhat_saved, ws, stream=0):
        """forward with the block epilogue y = act(layer(x) + skip) (include/sc_engine.h, sc_epilogue)"""
        ep = Epilogue(skip, preact, act, 0)
        self._check(self.lib.sc_layer_forward_ex(plan, byref(L), x, w, bias, byref(ep), y, xhat_saved, ws, stream))

    def transform_inverse_ex(self, plan, mode, yhat_ptr, bias_ptr, channels, skip, preact, act, y_ptr, n_images,
                             ws_ptr, stream=0):
        ep = Epilogue(skip, preact, act, 0)
        self._check(self.lib.sc_transform_inverse_ex(plan, mode, yhat_ptr, bias_ptr, channels, byref(ep), y_ptr,
                                                     n_images, ws_ptr, stream))

    def layer_backward(self, plan, L, gy, xhat_saved, w, gx, gw, gbias, ws, stream=0):
        self._check(self.lib.sc_layer_backward(plan, byref(L), gy, xhat_saved, w, gx, gw,
                                               gbias, ws, stream))

    def layer_backward_ex(self, plan, L, gy, xhat_saved, w, gx, gw, gbias, gx_addend, ws, stream=0):
        self._check(self.lib.sc_layer_backward_ex(plan, byref(L), gy, xhat_saved, w, gx, gw, gbias, gx_addend, ws, stream))


_LIB = None


def get_lib():
    """The product's engine library (HIP build); raises EngineError when it is not built."""
    global _LIB
    if _LIB is None:
        # SC_ENGINE_LIB: another BUILD of the same engine (scripts/build_variants.py: A-B measurements of compile-time
        # switches through the whole module stack); a missing file fails as loudly as a missing default library
        _LIB = ScEngineLib(os.environ.get("SC_ENGINE_LIB") or DEFAULT_LIB)
    return _LIB
