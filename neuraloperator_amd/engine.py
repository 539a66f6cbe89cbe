"""Host side of the engine above the C-ABI: plan cache + autograd Functions.

PyTorch is plumbing here (device memory, streams, autograd graph); every arithmetic step
of the hot path runs in libsc_engine.so (hand-written HIP, include/sc_engine.h).
"""
import atexit
import collections
import threading

import torch

from . import _lib
from .modes import kept_block

SC_PLAN_COMPLEX = _lib.SC_PLAN_COMPLEX
SC_PLAN_IO_BF16 = _lib.SC_PLAN_IO_BF16
_PLAN_LOCK = threading.Lock()
_PLANS = collections.OrderedDict()     # least recently used first; the key starts with the device index
MAX_CACHED_PLANS = 512                 # PER DEVICE.  Variable-resolution / incremental-mode training creates a plan per
                                       # (grid, kept modes, frequency map); a plan is a few KB .. MB of device tables
_RETIRED = collections.deque()         # (event, handle) of evicted plans: freed on a later cache MISS once the event
                                       # has passed -- never inline on the hit path (sc_plan_destroy = hipFree = a
                                       # device-wide synchronisation)
_NO_BF16_IO = set()       # (device, spatial, kept, norm, flags) the engine has no bfloat16-I/O kernels for


def _require_gpu(t, what="input"):
    if not t.is_cuda:
        raise RuntimeError(
            f"neuraloperator_amd: {what} is on {t.device}; the SpectralConv engine runs on "
            "MI355X (ROCm) devices only and has no CPU path -- move the module and its inputs "
            "to a 'cuda' device.")


def _freeze(freq):
    if freq is None:
        return None
    return tuple(None if f is None else tuple(f) for f in freq)


def get_plan(device, spatial, kept, fft_norm="forward", flags=0, freq=None, real_col=0):
    """Cached ``sc_plan`` for (device, spatial sizes, kept modes, norm, frequency maps)."""
    key = (device.index if device.index is not None else torch.cuda.current_device(),
           tuple(int(s) for s in spatial), tuple(int(k) for k in kept), fft_norm, flags,
           _freeze(freq), int(real_col))
    with _PLAN_LOCK:
        plan = _PLANS.get(key)
        if plan is None:
            lib = _lib.get_lib()
            with torch.cuda.device(key[0]):
                plan = lib.plan_create(key[1], key[2], fft_norm=fft_norm, flags=flags, freq=freq,
                                       real_col=real_col)
            _PLANS[key] = plan
            _evict_and_reap(key[0])
        else:
            _PLANS.move_to_end(key)
    return plan


def _evict_and_reap(dev_index):
    """Called under _PLAN_LOCK on a cache miss (plan creation has just allocated and copied, i.e. already paid a
    synchronising call).  Plans of `dev_index` beyond the per-device cap are RETIRED behind an event recorded on the
    current stream; retired plans whose event has completed are released (their handle frees the device tables once
    no autograd context references it any more)."""
    mine = [k for k in _PLANS if k[0] == dev_index]
    for k in mine[:max(len(mine) - MAX_CACHED_PLANS, 0)]:          # least recently used first
        ev = None
        if torch.cuda.is_available():
            with torch.cuda.device(dev_index):
                ev = torch.cuda.Event()
                ev.record()
        _RETIRED.append((ev, _PLANS.pop(k)))
    for _ in range(len(_RETIRED)):
        ev, handle = _RETIRED.popleft()
        if ev is None or ev.query():
            del handle
        else:
            _RETIRED.append((ev, handle))


def get_plan_bf16_io(device, spatial, kept, fft_norm, flags):
    """Plan whose real tensors are bfloat16 in memory (SC_PLAN_IO_BF16), or None where the engine only has
    float32 I/O for the shape (everything off the fused 2-D kernels): the caller then converts."""
    key = (device.index, tuple(int(s) for s in spatial), tuple(int(k) for k in kept), fft_norm, flags)
    if key in _NO_BF16_IO:
        return None
    try:
        return get_plan(device, spatial, kept, fft_norm, flags | SC_PLAN_IO_BF16)
    except _lib.EngineError as e:
        if "SC_PLAN_IO_BF16 is implemented" not in str(e):
            raise                       # out of memory, HIP errors, ...: not a property of the shape
        _NO_BF16_IO.add(key)
        return None


@atexit.register
def _destroy_plans():
    # device memory is reclaimed with the process: the handles are dropped without calling into the library
    # (_lib marks the shutdown first: atexit handlers run in reverse order of registration, _lib is imported earlier)
    _lib._SHUTDOWN = True
    _PLANS.clear()
    _RETIRED.clear()


def _stream():
    # the raw handle of torch's current stream: torch.cuda.current_stream().cuda_stream builds a Stream object per call
    # (~12 us of host time, four to a dozen calls per layer step)
    try:
        return torch._C._cuda_getCurrentRawStream(torch.cuda.current_device())
    except AttributeError:                                      # a torch build without the private accessor
        return torch.cuda.current_stream().cuda_stream


def _ws(nbytes, device):
    return torch.empty(max(int(nbytes), 1), dtype=torch.uint8, device=device)


class SpectralConvDenseFn(torch.autograd.Function):
    """y = irfftn(pad(einsum('bi..,io..->bo..', trunc(rfftn(x)), W))) + bias, one C-ABI call
    each way (sc_layer_forward / sc_layer_backward).

    Replaces /root/reference/neuralop/layers/spectral_convolution.py:443-568 and the
    autograd graph PyTorch derives from it."""

    @staticmethod
    def forward(ctx, x, weight, bias, n_modes_attr, max_n_modes_attr, fft_norm, flags):
        _require_gpu(x)
        _require_gpu(weight, "weight")
        lib = _lib.get_lib()
        x = x.contiguous()
        w = weight.detach()
        if w.dtype != torch.complex64:
            w = w.to(torch.complex64)
        w = w.contiguous()
        b, cin = x.shape[:2]
        cout = w.shape[1]
        if w.shape[0] != cin:
            raise ValueError(f"input has {cin} channels, weight expects {w.shape[0]}")
        spatial = list(x.shape[2:])
        kept, w_start = kept_block(spatial, n_modes_attr, max_n_modes_attr)
        # bfloat16 activations: the fused kernels read / write them as they are (fp32 arithmetic, y and
        # gx rounded once on the store); other shapes and dtypes are converted to float32 here
        plan = get_plan_bf16_io(x.device, spatial, kept, fft_norm, flags) if x.dtype == torch.bfloat16 else None
        io_dtype = torch.bfloat16 if plan is not None else torch.float32
        if plan is None:
            x = x.float()
            plan = get_plan(x.device, spatial, kept, fft_norm, flags)
        L = lib.layer_desc(b, cin, cout, list(w.shape[2:]), w_start)
        with torch.cuda.device(x.device):
            ws = _ws(lib.layer_workspace_bytes(plan, L), x.device)
            y = torch.empty((b, cout, *spatial), dtype=io_dtype, device=x.device)
            xhat = torch.empty((b, cin, *kept, 2), dtype=torch.float32, device=x.device)
            bias_flat = None
            if bias is not None:
                bias_flat = bias.detach().reshape(-1).float().contiguous()
            lib.layer_forward(plan, L, x.data_ptr(), torch.view_as_real(w).data_ptr(),
                              0 if bias_flat is None else bias_flat.data_ptr(),
                              y.data_ptr(), xhat.data_ptr(), ws.data_ptr(), _stream())
        ctx.save_for_backward(xhat, w)
        ctx.plan, ctx.L = plan, L
        ctx.io_dtype = io_dtype
        ctx.x_shape = tuple(x.shape)
        ctx.bias_shape = None if bias is None else tuple(bias.shape)
        ctx.w_shape = tuple(weight.shape)
        return y

    @staticmethod
    def backward(ctx, gy):
        lib = _lib.get_lib()
        xhat, w = ctx.saved_tensors
        need_x, need_w, need_b = ctx.needs_input_grad[:3]
        gy = gy.contiguous()
        if gy.dtype != ctx.io_dtype:
            gy = gy.to(ctx.io_dtype)
        dev = gy.device
        with torch.cuda.device(dev):
            ws = _ws(lib.layer_workspace_bytes(ctx.plan, ctx.L), dev)
            gx = torch.empty(ctx.x_shape, dtype=ctx.io_dtype, device=dev) if need_x else None
            gw = None
            if need_w:
                full = all(ctx.L.w_start[d] == 0 for d in range(len(ctx.w_shape) - 2)) and \
                    tuple(xhat.shape[2:-1]) == tuple(ctx.w_shape[2:])
                alloc = torch.empty if full else torch.zeros
                gw = alloc((*ctx.w_shape, 2), dtype=torch.float32, device=dev)
            gb = None
            if need_b and ctx.bias_shape is not None:
                gb = torch.empty(ctx.L.cout, dtype=torch.float32, device=dev)
            lib.layer_backward(ctx.plan, ctx.L, gy.data_ptr(), xhat.data_ptr(),
                               torch.view_as_real(w).data_ptr(),
                               0 if gx is None else gx.data_ptr(),
                               0 if gw is None else gw.data_ptr(),
                               0 if gb is None else gb.data_ptr(), ws.data_ptr(), _stream())
        gw_c = torch.view_as_complex(gw) if gw is not None else None
        gb_r = gb.reshape(ctx.bias_shape) if gb is not None else None
        return gx, gw_c, gb_r, None, None, None, None


class FourierLayerFn(torch.autograd.Function):
    """out = act(spectral_conv(x) + skip): the Fourier layer of an FNO block (neuralop/layers/fno_block.py:392-414:
    ``x_fno + x_skip_fno`` followed by the non-linearity) with the addition and the activation done in the store
    path of the inverse transform -- ``sc_layer_forward_ex``.  ``act``: "gelu" (exact, torch's
    default) or None.  Backward: the activation's derivative is one elementwise pass over the saved pre-activation
    (ATen's gelu_backward), its result is both the skip branch's gradient and the input of ``sc_layer_backward``."""

    @staticmethod
    def forward(ctx, x, weight, bias, skip, act, n_modes_attr, max_n_modes_attr, fft_norm, flags):
        _require_gpu(x)
        _require_gpu(weight, "weight")
        lib = _lib.get_lib()
        if act not in (None, "gelu"):
            raise ValueError(f"activation {act!r}: the fused epilogue knows 'gelu' and None")
        x = x.contiguous().float()
        skip = skip.contiguous().float()
        w = weight.detach().to(torch.complex64).contiguous()
        b, cin = x.shape[:2]
        cout = w.shape[1]
        spatial = list(x.shape[2:])
        if tuple(skip.shape) != (b, cout, *spatial):
            raise ValueError(f"skip has shape {tuple(skip.shape)}, the layer's output is {(b, cout, *spatial)}")
        kept, w_start = kept_block(spatial, n_modes_attr, max_n_modes_attr)
        plan = get_plan(x.device, spatial, kept, fft_norm, flags)
        L = lib.layer_desc(b, cin, cout, list(w.shape[2:]), w_start)
        need_pre = act == "gelu" and any(ctx.needs_input_grad[:4])
        with torch.cuda.device(x.device):
            ws = _ws(lib.layer_workspace_bytes(plan, L), x.device)
            y = torch.empty((b, cout, *spatial), dtype=torch.float32, device=x.device)
            pre = torch.empty_like(y) if need_pre else None
            xhat = torch.empty((b, cin, *kept, 2), dtype=torch.float32, device=x.device)
            bias_flat = None if bias is None else bias.detach().reshape(-1).float().contiguous()
            lib.layer_forward_ex(plan, L, x.data_ptr(), torch.view_as_real(w).data_ptr(),
                                 0 if bias_flat is None else bias_flat.data_ptr(), skip.data_ptr(),
                                 0 if pre is None else pre.data_ptr(),
                                 _lib.SC_ACT_GELU if act == "gelu" else _lib.SC_ACT_NONE,
                                 y.data_ptr(), xhat.data_ptr(), ws.data_ptr(), _stream())
        ctx.save_for_backward(xhat, w, pre)
        ctx.plan, ctx.L, ctx.act = plan, L, act
        ctx.x_shape, ctx.w_shape = tuple(x.shape), tuple(weight.shape)
        ctx.bias_shape = None if bias is None else tuple(bias.shape)
        return y

    @staticmethod
    def backward(ctx, gout):
        lib = _lib.get_lib()
        xhat, w, pre = ctx.saved_tensors
        need_x, need_w, need_b, need_s = ctx.needs_input_grad[:4]
        gz = gout.contiguous().float()
        if ctx.act == "gelu":
            gz = torch.ops.aten.gelu_backward(gz, pre)              # host glue: one elementwise pass
        dev = gz.device
        gx = gw = gb = None
        if need_x or need_w or (need_b and ctx.bias_shape is not None):
            with torch.cuda.device(dev):
                ws = _ws(lib.layer_workspace_bytes(ctx.plan, ctx.L), dev)
                gx = torch.empty(ctx.x_shape, dtype=torch.float32, device=dev) if need_x else None
                if need_w:
                    full = all(ctx.L.w_start[d] == 0 for d in range(len(ctx.w_shape) - 2)) and \
                        tuple(xhat.shape[2:-1]) == tuple(ctx.w_shape[2:])
                    gw = (torch.empty if full else torch.zeros)((*ctx.w_shape, 2), dtype=torch.float32, device=dev)
                if need_b and ctx.bias_shape is not None:
                    gb = torch.empty(ctx.L.cout, dtype=torch.float32, device=dev)
                lib.layer_backward(ctx.plan, ctx.L, gz.data_ptr(), xhat.data_ptr(), torch.view_as_real(w).data_ptr(),
                                   0 if gx is None else gx.data_ptr(), 0 if gw is None else gw.data_ptr(),
                                   0 if gb is None else gb.data_ptr(), ws.data_ptr(), _stream())
        return (gx, None if gw is None else torch.view_as_complex(gw),
                None if gb is None else gb.reshape(ctx.bias_shape), gz if need_s else None,
                None, None, None, None, None)


class TransformForwardFn(torch.autograd.Function):
    """x (B, C, d1..dN) real -> truncated spectrum (B, C, k1..kN) complex (weight order).
    forward = SC_FWD_SCALED, backward = its adjoint SC_INV_ADJ_R2C."""

    @staticmethod
    def forward(ctx, x, kept, fft_norm, flags, freq=None):
        _require_gpu(x)
        lib = _lib.get_lib()
        cplx = bool(flags & _lib.SC_PLAN_COMPLEX)
        if cplx:
            x = torch.view_as_real(x.contiguous().to(torch.complex64)).contiguous()
            spatial = list(x.shape[2:-1])
        else:
            x = x.contiguous().float()
            spatial = list(x.shape[2:])
        b, c = x.shape[:2]
        plan = get_plan(x.device, spatial, kept, fft_norm, flags, freq)
        with torch.cuda.device(x.device):
            ws = _ws(lib.plan_workspace_bytes(plan, b * c), x.device)
            xhat = torch.empty((b, c, *kept, 2), dtype=torch.float32, device=x.device)
            lib.transform_forward(plan, _lib.SC_FWD_SCALED, x.data_ptr(), xhat.data_ptr(), b * c,
                                  ws.data_ptr(), _stream())
        ctx.plan = plan
        ctx.x_shape = tuple(x.shape)
        ctx.cplx = cplx
        return torch.view_as_complex(xhat)

    @staticmethod
    def backward(ctx, gxhat):
        lib = _lib.get_lib()
        g = torch.view_as_real(gxhat.contiguous().to(torch.complex64)).contiguous()
        b, c = ctx.x_shape[:2]
        with torch.cuda.device(g.device):
            ws = _ws(lib.plan_workspace_bytes(ctx.plan, b * c), g.device)
            gx = torch.empty(ctx.x_shape, dtype=torch.float32, device=g.device)
            lib.transform_inverse(ctx.plan, _lib.SC_INV_ADJ_R2C, g.data_ptr(), 0, c, gx.data_ptr(),
                                  b * c, ws.data_ptr(), _stream())
        if ctx.cplx:
            gx = torch.view_as_complex(gx)
        return gx, None, None, None, None


class TransformInverseFn(torch.autograd.Function):
    """truncated spectrum (B, C, k..) complex -> y (B, C, d..) real (+ bias).
    forward = SC_INV_PADDED, backward = its adjoint SC_FWD_ADJ_C2R (bias grad from DC)."""

    @staticmethod
    def forward(ctx, yhat, bias, spatial, fft_norm, flags, freq=None, real_col=0):
        _require_gpu(yhat)
        lib = _lib.get_lib()
        cplx = bool(flags & _lib.SC_PLAN_COMPLEX)
        if cplx and bias is not None:
            raise ValueError("complex-data inverse transform takes no bias (add it afterwards)")
        yh = torch.view_as_real(yhat.contiguous().to(torch.complex64)).contiguous()
        b, c = yh.shape[:2]
        kept = list(yh.shape[2:-1])
        plan = get_plan(yh.device, spatial, kept, fft_norm, flags, freq, real_col)
        with torch.cuda.device(yh.device):
            ws = _ws(lib.plan_workspace_bytes(plan, b * c), yh.device)
            y = torch.empty((b, c, *spatial) + ((2,) if cplx else ()), dtype=torch.float32, device=yh.device)
            bias_flat = None if bias is None else bias.detach().reshape(-1).float().contiguous()
            lib.transform_inverse(plan, _lib.SC_INV_PADDED, yh.data_ptr(),
                                  0 if bias_flat is None else bias_flat.data_ptr(), c,
                                  y.data_ptr(), b * c, ws.data_ptr(), _stream())
        ctx.plan = plan
        ctx.kept = kept
        ctx.cplx = cplx
        ctx.bias_shape = None if bias is None else tuple(bias.shape)
        return torch.view_as_complex(y) if cplx else y

    @staticmethod
    def backward(ctx, gy):
        lib = _lib.get_lib()
        if ctx.cplx:
            gy = torch.view_as_real(gy.contiguous().to(torch.complex64)).contiguous()
        else:
            gy = gy.contiguous().float()
        b, c = gy.shape[:2]
        with torch.cuda.device(gy.device):
            ws = _ws(lib.plan_workspace_bytes(ctx.plan, b * c), gy.device)
            ghat = torch.empty((b, c, *ctx.kept, 2), dtype=torch.float32, device=gy.device)
            lib.transform_forward(ctx.plan, _lib.SC_FWD_ADJ_C2R, gy.data_ptr(), ghat.data_ptr(),
                                  b * c, ws.data_ptr(), _stream())
            gb = None
            if ctx.bias_shape is not None and ctx.needs_input_grad[1]:
                gb = torch.empty(c, dtype=torch.float32, device=gy.device)
                lib.bias_grad(ctx.plan, ghat.data_ptr(), b, c, gb.data_ptr(), _stream())
                gb = gb.reshape(ctx.bias_shape)
        return torch.view_as_complex(ghat), gb, None, None, None, None, None


def _cview(t):
    """complex64 contiguous tensor -> float32 view pointer holder."""
    t = t.to(torch.complex64) if t.dtype != torch.complex64 else t
    return torch.view_as_real(t.contiguous())


def modegemm(a, b, *, P, Q, R, n_modes, a_strides, b_strides, out, c_strides, conj_a=False,
             conj_b=False, b_idx=None, c_idx=None, accumulate=False):
    """Raw mode-batched complex GEMM (sc_modegemm).  ``a``, ``b``, ``out`` are complex64 CUDA
    tensors; strides are (sp, sr, sm) / (sr, sq, sm) / (sp, sq, sm) in complex elements."""
    lib = _lib.get_lib()
    av, bv, cv = torch.view_as_real(a), torch.view_as_real(b), torch.view_as_real(out)
    with torch.cuda.device(out.device):
        lib.modegemm(av.data_ptr(), bv.data_ptr(), cv.data_ptr(), _stream(),
                     P=P, Q=Q, R=R, n_modes=n_modes,
                     a_sp=a_strides[0], a_sr=a_strides[1], a_sm=a_strides[2],
                     b_sr=b_strides[0], b_sq=b_strides[1], b_sm=b_strides[2],
                     c_sp=c_strides[0], c_sq=c_strides[1], c_sm=c_strides[2],
                     conj_a=int(conj_a), conj_b=int(conj_b), accumulate=int(accumulate),
                     b_idx=0 if b_idx is None else b_idx.data_ptr(),
                     c_idx=0 if c_idx is None else c_idx.data_ptr())
    return out


def dense_contract_backward(xhat, w, g, need_x=True, need_w=True):
    """The two autograd einsums of 'bixy,ioxy->boxy' (spectral_convolution.py:21-46):
    gxhat[b,i,m] = sum_o g[b,o,m] conj(w[i,o,m]),  gw[i,o,m] = sum_b conj(xhat[b,i,m]) g[b,o,m]
    on contiguous (b, c, modes...) complex64 blocks.  When both are wanted they go down as ONE call
    (sc_modegemm_pair: one launch of k_modegemm_dma_bwd when the pair qualifies, else two launches)."""
    b, ci = xhat.shape[:2]
    co = w.shape[1]
    mk = xhat[0, 0].numel()
    gx = gw = None
    if need_x and need_w and b and ci and co and mk:
        kw_w = dict(P=ci, Q=co, R=b, n_modes=mk, a_sp=mk, a_sr=ci * mk, a_sm=1, conj_a=1,
                    b_sr=co * mk, b_sq=mk, b_sm=1, c_sp=co * mk, c_sq=mk, c_sm=1)
        kw_x = dict(P=b, Q=ci, R=co, n_modes=mk, a_sp=co * mk, a_sr=mk, a_sm=1,
                    b_sr=mk, b_sq=co * mk, b_sm=1, conj_b=1, c_sp=ci * mk, c_sq=mk, c_sm=1)
        gx, gw = torch.empty_like(xhat), torch.empty_like(w)
        ptr = lambda t: torch.view_as_real(t).data_ptr()
        with torch.cuda.device(g.device):
            _lib.get_lib().modegemm_pair(kw_w, ptr(xhat), ptr(g), ptr(gw), kw_x, ptr(g), ptr(w), ptr(gx), _stream())
        return gx, gw
    if need_x:
        gx = torch.empty_like(xhat)
        modegemm(g, w, P=b, Q=ci, R=co, n_modes=mk, a_strides=(co * mk, mk, 1), b_strides=(mk, co * mk, 1),
                 conj_b=True, out=gx, c_strides=(ci * mk, mk, 1))
    if need_w:
        gw = torch.empty_like(w)
        modegemm(xhat, g, P=ci, Q=co, R=b, n_modes=mk, a_strides=(mk, ci * mk, 1), conj_a=True,
                 b_strides=(co * mk, mk, 1), out=gw, c_strides=(co * mk, mk, 1))
    return gx, gw


class ModeContractDenseFn(torch.autograd.Function):
    """yhat[b,o,m] = sum_i xhat[b,i,m] * w[i,o,m] on an arbitrary (e.g. mode-sharded) block of
    modes; w has exactly the mode extents of xhat.  Three sc_modegemm launches in total
    (forward, gX, gW) -- the einsum 'bixy,ioxy->boxy' of spectral_convolution.py:21-46 and its
    two autograd einsums."""

    @staticmethod
    def forward(ctx, xhat, w):
        _require_gpu(xhat, "xhat")
        xhat = xhat.contiguous().to(torch.complex64)
        w = w.contiguous().to(torch.complex64)
        b, ci = xhat.shape[:2]
        co = w.shape[1]
        if tuple(w.shape[2:]) != tuple(xhat.shape[2:]) or w.shape[0] != ci:
            raise ValueError(f"weight block {tuple(w.shape)} does not match spectrum {tuple(xhat.shape)}")
        mk = 1
        for k in xhat.shape[2:]:
            mk *= int(k)
        yhat = torch.empty((b, co, *xhat.shape[2:]), dtype=torch.complex64, device=xhat.device)
        modegemm(xhat, w, P=b, Q=co, R=ci, n_modes=mk, a_strides=(ci * mk, mk, 1),
                 b_strides=(co * mk, mk, 1), out=yhat, c_strides=(co * mk, mk, 1))
        ctx.save_for_backward(xhat, w)
        ctx.dims = (b, ci, co, mk)
        return yhat

    @staticmethod
    def backward(ctx, g):
        xhat, w = ctx.saved_tensors
        b, ci, co, mk = ctx.dims
        g = g.contiguous().to(torch.complex64)
        gx, gw = dense_contract_backward(xhat, w, g, ctx.needs_input_grad[0], ctx.needs_input_grad[1])
        return gx, gw


# ------------------------------------------------------------------------------------------
# generic mode GEMM with autograd: the pairwise steps of the factorized contractions
# ------------------------------------------------------------------------------------------
# row count from which the gradient steps of ModeGemmFn re-label a long reduction onto the lanes
# (tests set it to 1 to force those paths at fixture sizes)
RELABEL_MIN_ROWS = 4096


def _strides3(t, lead):
    """(s0, s1, sm) in complex elements of a [lead0, lead1, M] or [lead0, lead1] (mode-independent) view."""
    if t.dim() == 3:
        return int(t.stride(0)), int(t.stride(1)), int(t.stride(2))
    return int(t.stride(0)), int(t.stride(1)), 0


def _raw_mode_gemm(a, b, n_modes, conj_a, conj_b, reduce_modes=False, flags=0):
    """a: [P, R, M] or [P, R]; b: [R, Q, M] or [R, Q]; complex64 CUDA tensors/views (any strides).
    Returns C[P, Q, M] (or C[P, Q] = sum over modes when reduce_modes)."""
    lib = _lib.get_lib()
    # the lanes run over the modes: an operand whose mode stride is not 1 would be gathered
    if a.dim() == 3 and a.stride(2) != 1 and a.shape[2] > 1:
        a = a.contiguous()
    if b.dim() == 3 and b.stride(2) != 1 and b.shape[2] > 1:
        b = b.contiguous()
    P, R = int(a.shape[0]), int(a.shape[1])
    Q = int(b.shape[1])
    a_sp, a_sr, a_sm = _strides3(a, 2)
    b_sr, b_sq, b_sm = _strides3(b, 2)
    dev = a.device
    with torch.cuda.device(dev):
        if reduce_modes:
            kw = dict(P=P, Q=Q, R=R, n_modes=n_modes, a_sp=a_sp, a_sr=a_sr, a_sm=a_sm, b_sr=b_sr, b_sq=b_sq, b_sm=b_sm,
                      conj_a=int(conj_a), conj_b=int(conj_b), flags=int(flags), c_sp=Q, c_sq=1, c_sm=0)
            ws_bytes = lib.modegemm_msum_workspace_bytes(**kw) if P and Q and n_modes and R else 0
            if ws_bytes:
                # matrix-core kernel + fixed-order reduction (sc_kernels_fmx.h): C is overwritten, no zero fill
                out = torch.empty((P, Q), dtype=torch.complex64, device=dev)
                ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
                lib.modegemm_msum_ws(a.data_ptr(), b.data_ptr(), out.data_ptr(), ws.data_ptr(), ws_bytes, _stream(), **kw)
                return out
            out = torch.zeros((P, Q), dtype=torch.complex64, device=dev)
            fn, c = lib.modegemm_msum, dict(c_sp=Q, c_sq=1, c_sm=0)
        else:
            out = torch.empty((P, Q, n_modes), dtype=torch.complex64, device=dev)
            fn, c = lib.modegemm, dict(c_sp=Q * n_modes, c_sq=n_modes, c_sm=1)
        if P and Q and n_modes:
            fn(a.data_ptr(), b.data_ptr(), out.data_ptr(), _stream(), P=P, Q=Q, R=R, n_modes=n_modes,
               a_sp=a_sp, a_sr=a_sr, a_sm=a_sm, b_sr=b_sr, b_sq=b_sq, b_sm=b_sm,
               conj_a=int(conj_a), conj_b=int(conj_b), flags=int(flags), **c)
    return out


class ModeGemmFn(torch.autograd.Function):
    """C[p,q,m] = sum_r opA(A)[p,r,m] * opB(B)[r,q,m] with either operand possibly mode-independent
    (a 2-D tensor: Tucker / CP factor matrices).  Forward and both gradients run on sc_modegemm /
    sc_modegemm_msum (gradient of a mode-independent operand = sum over the modes: wave reduction).
    One pairwise step of the reference's _contract_tucker / _contract_cp einsums
    (spectral_convolution.py:55-103) and of their autograd."""

    @staticmethod
    def forward(ctx, a, b, n_modes, conj_a, conj_b, flags=0):
        _require_gpu(a, "A")
        _require_gpu(b, "B")
        a = a if a.dtype == torch.complex64 else a.to(torch.complex64)
        b = b if b.dtype == torch.complex64 else b.to(torch.complex64)
        if a.shape[1] != b.shape[0]:
            raise ValueError(f"inner extents differ: A {tuple(a.shape)} B {tuple(b.shape)}")
        ctx.save_for_backward(a, b)
        ctx.cfg = (int(n_modes), bool(conj_a), bool(conj_b))
        ctx.flags = int(flags)                # SC_GEMM_F16: the complex-half contraction, forward AND gradients
        if ctx.flags & _lib.SC_GEMM_F16 and (a.dim() != 3 or b.dim() != 3):
            raise ValueError("SC_GEMM_F16 contracts two per-mode operands (the dense equation)")
        return _raw_mode_gemm(a, b, int(n_modes), conj_a, conj_b, flags=ctx.flags)

    @staticmethod
    def backward(ctx, gc):
        a, b = ctx.saved_tensors
        M, ca, cb = ctx.cfg
        gc = gc.contiguous()
        P, R, Q = int(a.shape[0]), int(a.shape[1]), int(b.shape[1])
        ga = gb = None
        if ctx.needs_input_grad[0]:
            # grad_a = sum_q gC * conj(opB(B)); conj_a: grad_A = conj(grad_a) = sum_q conj(gC) * opB(B)
            cbx = cb if ca else not cb
            if a.dim() == 2 and b.dim() == 3 and P >= RELABEL_MIN_ROWS and M < 64:
                # A is mode independent and tall, the modes are few (core x last-mode-factor step of a Tucker
                # weight: 46656 x 19 against 33 modes): a plain (P x QM)(QM x R) product with the ROWS on the
                # lanes instead of a 33-lane wave reduction per row -- 128 -> ~25 us
                gt = gc.reshape(P, Q * M).t().contiguous().unsqueeze(0)                # [1, QM, P]
                bm = b.permute(1, 2, 0).reshape(Q * M, R)                              # [QM, R], mode independent
                ga = _raw_mode_gemm(gt, bm, P, ca, cbx).reshape(R, P).t()
            else:
                ga = _raw_mode_gemm(gc, b.transpose(0, 1), M, ca, cbx, reduce_modes=(a.dim() == 2), flags=ctx.flags)
        if ctx.needs_input_grad[1]:
            # grad_b = sum_p conj(opA(A)) * gC; conj_b: grad_B = conj(grad_b) = sum_p opA(A) * conj(gC)
            cax = ca if cb else not ca
            if a.dim() == 2 and b.dim() == 3 and P >= 8 * Q * M:
                # long reduction over p, few outputs (core x last-mode-factor step of a Tucker weight):
                # put p on the lanes -- sc_modegemm_msum with modes := p, columns := (q, m) -- instead
                # of a handful of workgroups walking all of p serially (21.7 ms -> ~0.1 ms at rank 0.1);
                # both operands are laid out with p contiguous first (a strided view gathers: 296 -> ~30 us)
                a2 = a.t().contiguous().unsqueeze(1)                                    # [R, 1, P]
                g2 = gc.reshape(P, Q * M).t().contiguous().unsqueeze(0)                 # [1, Q M, P]
                gb = _raw_mode_gemm(a2, g2, P, cax, cb, reduce_modes=True).reshape(R, Q, M)
            elif b.dim() == 2 and a.dim() == 3 and P * M >= RELABEL_MIN_ROWS and P >= 8 * R:
                # B is mode independent and the sum runs over many rows p AND the modes (mode-factor step:
                # 1296 rows x 33 modes -> a 36 x 64 gradient): (p, m) jointly on the lanes, 430 -> ~40 us
                a2 = a.permute(1, 0, 2).reshape(R, 1, P * M)                            # [R, 1, (p m)] (copy)
                g2 = gc.permute(1, 0, 2).reshape(1, Q, P * M)                           # [1, Q, (p m)] (copy)
                gb = _raw_mode_gemm(a2, g2, P * M, cax, cb, reduce_modes=True)
            else:
                gb = _raw_mode_gemm(a.transpose(0, 1), gc, M, cax, cb, reduce_modes=(b.dim() == 2), flags=ctx.flags)
        return ga, gb, None, None, None, None


class TuckerModes2dFn(torch.autograd.Function):
    """T[f, g, x, y] = sum_{c, d} core[f, g, c, d] U_x[x, c] U_y[y, d] and its three gradients, one launch each way
    (sc_tucker_modes_forward / _backward): the batch-independent part of the 2-D Tucker contraction
    (spectral_convolution.py:76-103)."""

    @staticmethod
    def forward(ctx, core, ux, uy):
        _require_gpu(core, "core")
        lib = _lib.get_lib()
        f, g, rx, ry = (int(v) for v in core.shape)
        mx, my = int(ux.shape[0]), int(uy.shape[0])
        c, a, b = (t.detach().to(torch.complex64).contiguous() for t in (core, ux, uy))
        out = torch.empty((f, g, mx * my), dtype=torch.complex64, device=core.device)
        with torch.cuda.device(core.device):
            lib.tucker_modes_forward(f * g, rx, ry, mx, my, torch.view_as_real(c).data_ptr(), torch.view_as_real(a).data_ptr(),
                                     torch.view_as_real(b).data_ptr(), torch.view_as_real(out).data_ptr(), _stream())
        ctx.save_for_backward(c, a, b)
        return out

    @staticmethod
    def backward(ctx, gt):
        c, a, b = ctx.saved_tensors
        lib = _lib.get_lib()
        f, g, rx, ry = (int(v) for v in c.shape)
        mx, my = int(a.shape[0]), int(b.shape[0])
        gt = gt.to(torch.complex64).contiguous()
        gc, ga, gb = torch.empty_like(c), torch.empty_like(a), torch.empty_like(b)
        ws = torch.empty(lib.tucker_modes_workspace_bytes(f * g, rx, ry, mx, my), dtype=torch.uint8, device=c.device)
        p = lambda t: torch.view_as_real(t).data_ptr()
        with torch.cuda.device(c.device):
            lib.tucker_modes_backward(f * g, rx, ry, mx, my, p(c), p(a), p(b), p(gt), p(gc), p(ga), p(gb), ws.data_ptr(),
                                      _stream())
        return gc, ga, gb


def tucker_modes_2d(core, ux, uy):
    """None when the sizes are outside the kernel's limits (the caller then takes the chain of mode GEMMs)."""
    f, g, rx, ry = (int(v) for v in core.shape)
    if not _lib.get_lib().tucker_modes_supported(f * g, rx, ry, int(ux.shape[0]), int(uy.shape[0])):
        return None
    return TuckerModes2dFn.apply(core, ux, uy)


class TuckerChainFn(torch.autograd.Function):
    """The activation side of the factorized Tucker contraction as ONE autograd node (round 3):
        z[b,f,m] = sum_i xhat[b,i,m] U_in[i,f];  t[b,g,m] = sum_f z[b,f,m] T[f,g,m];  yhat[b,o,m] = sum_g t[b,g,m] U_out[o,g]
    (the pairwise order of _forward_tucker, spectral_convolution.py:76-103) -- the same launches as three ModeGemmFn
    nodes, but a step of this layer is ~30 launches of 20-100 us each and the host needs about as long to walk four
    autograd nodes and their views as the device needs to run them (scripts/tfno_cpu_bound.py: 0.62 ms to issue a
    0.77 ms step).  Session 2: ONE C-ABI call per direction (sc_tucker_chain_forward / _backward) issues the three /
    six products from C++.  Backward: the six products of the three nodes, in the order their gradients are needed."""

    @staticmethod
    def forward(ctx, xhat, u_in, t3, u_out):
        _require_gpu(xhat, "xhat")
        c64 = lambda v: (v if v.dtype == torch.complex64 else v.to(torch.complex64)).contiguous()
        xhat, u_in, t3, u_out = c64(xhat), c64(u_in), c64(t3), c64(u_out)
        b, ci, m = (int(v) for v in xhat.shape)
        r1, r2, co = int(u_in.shape[1]), int(u_out.shape[1]), int(u_out.shape[0])
        if tuple(t3.shape) != (r1, r2, m) or int(u_in.shape[0]) != ci:
            raise ValueError(f"tucker_chain: xhat {tuple(xhat.shape)} u_in {tuple(u_in.shape)} t3 {tuple(t3.shape)} "
                             f"u_out {tuple(u_out.shape)} do not chain")
        dev = xhat.device
        z = torch.empty((b, r1, m), dtype=torch.complex64, device=dev)
        t = torch.empty((b, r2, m), dtype=torch.complex64, device=dev)
        yhat = torch.empty((b, co, m), dtype=torch.complex64, device=dev)
        ctx.dims = (b, ci, co, r1, r2, m)
        ctx.fused = False
        t3m = None
        if b and m:
            lib = _lib.get_lib()
            with torch.cuda.device(dev):
                # round 5: ONE launch for the three products where the shape fits (batch <= 32, <= 64 channels, ranks <= 48,
                # extents multiples of 4 -- BASELINE configs[2]): a workgroup walks xhat -> z -> t -> yhat for four modes out of
                # LDS (csrc/sc_kernels_tkchain.h); t3m = the mode-major copy of t3 that call writes and the backward call reads
                ctx.fused = lib.tucker_chain_fused_supported(ctx.dims) and all(
                    v.data_ptr() % 16 == 0 for v in (xhat, z, t, yhat))
                if ctx.fused:
                    t3m = torch.empty((m, r1, r2), dtype=torch.complex64, device=dev)
                    lib.tucker_chain_forward_fused(ctx.dims, xhat.data_ptr(), u_in.data_ptr(), t3.data_ptr(), u_out.data_ptr(),
                                                   t3m.data_ptr(), z.data_ptr(), t.data_ptr(), yhat.data_ptr(), _stream())
                else:
                    # ONE host call for the three products (round 3, session 2: sc_tucker_chain_forward issues the same three
                    # sc_modegemm launches from C++; a factorized step was ~0.6-0.76 ms of interpreter time per 0.76 ms of
                    # device time)
                    lib.tucker_chain_forward(ctx.dims, xhat.data_ptr(), u_in.data_ptr(), t3.data_ptr(), u_out.data_ptr(),
                                             z.data_ptr(), t.data_ptr(), yhat.data_ptr(), _stream())
        if t3m is not None:
            ctx.save_for_backward(xhat, u_in, t3, u_out, z, t, t3m)
        else:
            ctx.save_for_backward(xhat, u_in, t3, u_out, z, t)
        return yhat

    @staticmethod
    def backward(ctx, gy):
        xhat, u_in, t3, u_out, z, t = ctx.saved_tensors[:6]
        b, ci, co, r1, r2, m = ctx.dims
        gy = (gy if gy.dtype == torch.complex64 else gy.to(torch.complex64)).contiguous()
        need = ctx.needs_input_grad
        dev = xhat.device
        new = lambda *sh: torch.empty(sh, dtype=torch.complex64, device=dev)
        gx = new(b, ci, m) if need[0] else None
        gu_in = new(ci, r1) if need[1] else None
        gt3 = new(r1, r2, m) if need[2] else None
        gu_out = new(co, r2) if need[3] else None
        if b and m:
            lib = _lib.get_lib()
            p = lambda v: 0 if v is None else v.data_ptr()
            if ctx.fused and gy.data_ptr() % 16 == 0:
                # one launch: gy -> gt -> gz -> gxhat per four-mode tile, the t3 gradient mode-major (transposed back by the
                # call), the two factor gradients as per-workgroup partial sums reduced in fixed order.  The kernel always
                # forms the t3 gradient, so a scratch tensor stands in when autograd does not ask for it.
                t3m = ctx.saved_tensors[6]
                with torch.cuda.device(dev):
                    nb = lib.tucker_chain_backward_fused_workspace_bytes(ctx.dims)
                    ws = torch.empty(nb, dtype=torch.uint8, device=dev)
                    gt3_out = gt3 if gt3 is not None else new(r1, r2, m)
                    lib.tucker_chain_backward_fused(ctx.dims, xhat.data_ptr(), u_in.data_ptr(), t3m.data_ptr(), u_out.data_ptr(),
                                                    z.data_ptr(), t.data_ptr(), gy.data_ptr(), p(gx), p(gu_in),
                                                    gt3_out.data_ptr(), p(gu_out), ws.data_ptr(), nb, _stream())
                return gx, gu_in, gt3, gu_out
            with torch.cuda.device(dev):
                nb = lib.tucker_chain_workspace_bytes(ctx.dims)
                ws = torch.empty(nb, dtype=torch.uint8, device=dev)
                # the six products of the three steps, in the order their gradients are needed: gt = gy conj(U_out),
                # gU_out = sum t^H gy, gz = gt T^H, gT = z^H gt, gxhat = gz U_in^H, gU_in = sum xhat^H gz
                lib.tucker_chain_backward(ctx.dims, xhat.data_ptr(), u_in.data_ptr(), t3.data_ptr(), u_out.data_ptr(),
                                          z.data_ptr(), t.data_ptr(), gy.data_ptr(), p(gx), p(gu_in), p(gt3), p(gu_out),
                                          ws.data_ptr(), nb, _stream())
        else:
            for v in (gx, gu_in, gt3, gu_out):
                if v is not None:
                    v.zero_()
        return gx, gu_in, gt3, gu_out


def tucker_chain(xhat, u_in, t3, u_out):
    return TuckerChainFn.apply(xhat, u_in, t3, u_out)


def mode_gemm(a, b, n_modes, conj_a=False, conj_b=False, flags=0):
    return ModeGemmFn.apply(a, b, n_modes, conj_a, conj_b, flags)


class RoundF16Fn(torch.autograd.Function):
    """``t.half()`` of the reference's half / mixed precision modes (spectral_convolution.py:436-437 and the float16
    result of the inverse transform of a complex32 spectrum) with the values kept in fp32 storage: sc_round_f16.
    The cast's autograd is a cast back, i.e. the gradient passes through."""

    @staticmethod
    def forward(ctx, t):
        _require_gpu(t, "t")
        t = t.contiguous()
        out = torch.empty_like(t)
        with torch.cuda.device(t.device):
            _lib.get_lib().round_f16(t.data_ptr(), out.data_ptr(), t.numel(), _stream())
        return out

    @staticmethod
    def backward(ctx, g):
        return g


def round_f16(t):
    return RoundF16Fn.apply(t)


class EngineOps:
    """The three local stages of a (mode-parallel) spectral layer on the MI355X engine."""

    def __init__(self, fft_norm="forward", flags=0):
        self.fft_norm, self.flags = fft_norm, flags

    def forward_transform(self, x, kept, freq=None):
        return TransformForwardFn.apply(x, list(kept), self.fft_norm, self.flags, freq)

    def contract(self, xhat, w):
        return ModeContractDenseFn.apply(xhat, w)

    def inverse_transform(self, yhat, bias, spatial, freq=None, real_col=0):
        return TransformInverseFn.apply(yhat, bias, list(spatial), self.fft_norm, self.flags, freq, real_col)

    @staticmethod
    def contract_separable(xhat, w):
        """yhat[b, c, m] = xhat[b, c, m] w[c, m] (spectral_convolution.py:49-52): one 1 x 1 product per (channel, mode)"""
        b, c = int(xhat.shape[0]), int(xhat.shape[1])
        m = c
        for k in xhat.shape[2:]:
            m *= int(k)
        return mode_gemm(xhat.reshape(b, 1, m), w.reshape(1, 1, m), m).reshape(xhat.shape)

    # one complex axis of a separable transform (the sharded dim of mpu.SpatialParallelSpectralConv):
    # x (B, L, n) complex -> (B, L, k) with kept row r reading FFT index rows[r], and its zero-padded inverse;
    # 1-d complex plans with an explicit frequency map (sc_plan_desc.freq)
    def forward_axis(self, x, k, rows):
        return TransformForwardFn.apply(x, [int(k)], self.fft_norm, self.flags | SC_PLAN_COMPLEX, [list(rows)])

    def inverse_axis(self, xhat, n, rows):
        return TransformInverseFn.apply(xhat, None, [int(n)], self.fft_norm, self.flags | SC_PLAN_COMPLEX,
                                        [list(rows)], 0)


class EngineRawOps:
    """The local stages of a spectral layer and their adjoints as plain calls (no autograd): what a hand-scheduled
    pipeline (mpu.ModeParallelSpectralConv: transform chunk j+1 while chunk j is on the wire) is built from.
    ``out`` = a contiguous slice of a preallocated result the call writes into (no concatenation copies of the
    0.5 GB real tensors).  Real data, unchanged grid."""

    def __init__(self, fft_norm="forward", flags=0):
        self.fft_norm, self.flags = fft_norm, flags

    def _plan(self, dev, spatial, kept):
        return get_plan(dev, spatial, kept, self.fft_norm, self.flags)

    @staticmethod
    def _out(out, shape, dev):
        if out is None:
            return torch.empty(shape, dtype=torch.float32, device=dev)
        if tuple(out.shape) != tuple(shape) or not out.is_contiguous() or out.dtype != torch.float32:
            raise ValueError(f"out must be a contiguous float32 tensor of shape {tuple(shape)}")
        return out

    def fwd(self, x, kept):                                   # SC_FWD_SCALED
        _require_gpu(x)
        lib = _lib.get_lib()
        x = x.contiguous().float()
        n, c = x.shape[:2]
        plan = self._plan(x.device, list(x.shape[2:]), kept)
        with torch.cuda.device(x.device):
            ws = _ws(lib.plan_workspace_bytes(plan, n * c), x.device)
            xh = torch.empty((n, c, *kept, 2), dtype=torch.float32, device=x.device)
            lib.transform_forward(plan, _lib.SC_FWD_SCALED, x.data_ptr(), xh.data_ptr(), n * c, ws.data_ptr(), _stream())
        return torch.view_as_complex(xh)

    def fwd_adjoint(self, gxhat, spatial, out=None):          # SC_INV_ADJ_R2C
        lib = _lib.get_lib()
        g = _cview(gxhat)
        n, c = g.shape[:2]
        kept = list(g.shape[2:-1])
        plan = self._plan(g.device, list(spatial), kept)
        with torch.cuda.device(g.device):
            gx = self._out(out, (n, c, *spatial), g.device)
            ws = _ws(lib.plan_workspace_bytes(plan, n * c), g.device)
            lib.transform_inverse(plan, _lib.SC_INV_ADJ_R2C, g.data_ptr(), 0, c, gx.data_ptr(), n * c, ws.data_ptr(),
                                  _stream())
        return gx

    def inv(self, yhat, bias, spatial, out=None):             # SC_INV_PADDED (+ bias)
        lib = _lib.get_lib()
        yh = _cview(yhat)
        n, c = yh.shape[:2]
        kept = list(yh.shape[2:-1])
        plan = self._plan(yh.device, list(spatial), kept)
        with torch.cuda.device(yh.device):
            y = self._out(out, (n, c, *spatial), yh.device)
            ws = _ws(lib.plan_workspace_bytes(plan, n * c), yh.device)
            bflat = None if bias is None else bias.detach().reshape(-1).float().contiguous()
            lib.transform_inverse(plan, _lib.SC_INV_PADDED, yh.data_ptr(), 0 if bflat is None else bflat.data_ptr(), c,
                                  y.data_ptr(), n * c, ws.data_ptr(), _stream())
        return y

    def inv_adjoint(self, gy, kept, want_bias=False):         # SC_FWD_ADJ_C2R (+ bias gradient off the DC row)
        lib = _lib.get_lib()
        gy = gy.contiguous().float()
        n, c = gy.shape[:2]
        plan = self._plan(gy.device, list(gy.shape[2:]), kept)
        with torch.cuda.device(gy.device):
            ws = _ws(lib.plan_workspace_bytes(plan, n * c), gy.device)
            gh = torch.empty((n, c, *kept, 2), dtype=torch.float32, device=gy.device)
            lib.transform_forward(plan, _lib.SC_FWD_ADJ_C2R, gy.data_ptr(), gh.data_ptr(), n * c, ws.data_ptr(), _stream())
            gb = None
            if want_bias:
                gb = torch.empty(c, dtype=torch.float32, device=gy.device)
                lib.bias_grad(plan, gh.data_ptr(), n, c, gb.data_ptr(), _stream())
        return torch.view_as_complex(gh), gb

    # ---- the same four stages with the spectrum in the rank-major all-to-all layout [P][n][c][rows][rest]
    #      (include/sc_engine.h, sc_spectrum_shards): the buffer handed to / received from all_to_all_single is
    #      written / read in place, no permutation copy on either side of the collective
    def _sharded(self, lib, plan, n_images, P, rows, rest, dev):
        sh = lib.shards(P, rows, n_images * rows * rest)
        return sh, _ws(lib.plan_workspace_bytes_sharded(plan, n_images), dev)

    def fwd_sharded(self, x, kept, P, rows, out=None, mode=None):       # -> [P, n, c, rows, *kept[1:], 2] float32
        _require_gpu(x)
        lib = _lib.get_lib()
        x = x.contiguous().float()
        n, c = x.shape[:2]
        rest = 1
        for k in kept[1:]:
            rest *= int(k)
        plan = self._plan(x.device, list(x.shape[2:]), kept)
        with torch.cuda.device(x.device):
            buf = self._out(out, (P, n, c, rows, *kept[1:], 2), x.device)
            sh, ws = self._sharded(lib, plan, n * c, P, rows, rest, x.device)
            lib.transform_forward_sharded(plan, _lib.SC_FWD_SCALED if mode is None else mode, x.data_ptr(),
                                          buf.data_ptr(), n * c, sh, ws.data_ptr(), _stream())
        return buf

    def inv_adjoint_sharded(self, gy, kept, P, rows, out=None, want_bias=False):
        buf = self.fwd_sharded(gy, kept, P, rows, out=out, mode=_lib.SC_FWD_ADJ_C2R)
        gb = None
        if want_bias:
            lib = _lib.get_lib()
            n, c = gy.shape[:2]
            rest = 1
            for k in kept[1:]:
                rest *= int(k)
            plan = self._plan(gy.device, list(gy.shape[2:]), kept)
            with torch.cuda.device(gy.device):
                gb = torch.empty(c, dtype=torch.float32, device=gy.device)
                lib.bias_grad_sharded(plan, buf.data_ptr(), n, c, lib.shards(P, rows, n * c * rows * rest),
                                      gb.data_ptr(), _stream())
        return buf, gb

    def inv_sharded(self, buf, bias, spatial, k1, out=None, mode=None):   # buf [P, n, c, rows, rest.., 2] float32
        lib = _lib.get_lib()
        if buf.dtype != torch.float32 or not buf.is_contiguous():
            raise ValueError("inv_sharded: a contiguous float32 [P, n, c, rows, ..., 2] buffer is required")
        P, n, c, rows = (int(v) for v in buf.shape[:4])
        kept = [int(k1)] + [int(v) for v in buf.shape[4:-1]]
        rest = 1
        for k in kept[1:]:
            rest *= k
        plan = self._plan(buf.device, list(spatial), kept)
        with torch.cuda.device(buf.device):
            y = self._out(out, (n, c, *spatial), buf.device)
            sh, ws = self._sharded(lib, plan, n * c, P, rows, rest, buf.device)
            bflat = None if bias is None else bias.detach().reshape(-1).float().contiguous()
            lib.transform_inverse_sharded(plan, _lib.SC_INV_PADDED if mode is None else mode, buf.data_ptr(),
                                          0 if bflat is None else bflat.data_ptr(), c, y.data_ptr(), n * c, sh,
                                          ws.data_ptr(), _stream())
        return y

    def fwd_adjoint_sharded(self, buf, spatial, k1, out=None):
        return self.inv_sharded(buf, None, spatial, k1, out=out, mode=_lib.SC_INV_ADJ_R2C)

    @staticmethod
    def contract(xhat, w):
        b, ci = xhat.shape[:2]
        co = w.shape[1]
        mk = xhat[0, 0].numel()
        yhat = torch.empty((b, co, *xhat.shape[2:]), dtype=torch.complex64, device=xhat.device)
        modegemm(xhat, w, P=b, Q=co, R=ci, n_modes=mk, a_strides=(ci * mk, mk, 1), b_strides=(co * mk, mk, 1),
                 out=yhat, c_strides=(co * mk, mk, 1))
        return yhat

    @staticmethod
    def contract_bwd(xhat, w, ghat, need_x=True, need_w=True):
        return dense_contract_backward(xhat, w, ghat, need_x, need_w)


    @staticmethod
    def contract_separable(xhat, w):
        """yhat[b, c, m] = xhat[b, c, m] w[c, m] (``_contract_dense_separable``, spectral_convolution.py:49-52):
        one sc_modegemm launch with (channel, mode) as the lane index, R = Q = 1."""
        b = xhat.shape[0]
        m = w.numel()
        return _raw_mode_gemm(xhat.reshape(b, 1, m), w.reshape(1, 1, m), m, False, False).reshape(xhat.shape)

    @staticmethod
    def contract_separable_bwd(xhat, w, ghat, need_x=True, need_w=True):
        b = xhat.shape[0]
        m = w.numel()
        gx = _raw_mode_gemm(ghat.reshape(b, 1, m), w.reshape(1, 1, m), m, False, True).reshape(xhat.shape) \
            if need_x else None
        gw = _raw_mode_gemm(xhat.reshape(1, b, m), ghat.reshape(b, 1, m), m, True, False).reshape(w.shape) \
            if need_w else None
        return gx, gw

    def cp_dense(self, weights, factors):
        """Dense (Cin, Cout, modes...) block of a CP weight on the engine with autograd (SpectralConv._cp_dense)."""
        from types import SimpleNamespace
        from .spectral_conv import SpectralConv
        kept = [int(f.shape[0]) for f in factors[2:]]
        return SpectralConv._cp_dense(SpectralConv, SimpleNamespace(weights=weights, factors=list(factors)), kept)

    def tt_dense(self, cores):
        """Dense block of a tensor-train weight on the engine with autograd (SpectralConv._tt_dense)."""
        from types import SimpleNamespace
        from .spectral_conv import SpectralConv
        kept = [int(c.shape[1]) for c in cores[2:]]
        return SpectralConv._tt_dense(SimpleNamespace(factors=list(cores)), kept)

    def tucker_dense(self, core, factors):
        """Dense (Cin, Cout, modes...) block of a Tucker weight on the engine, with autograd to the core and every
        factor (the chain of sc_modegemm launches of SpectralConv._tucker_dense)."""
        from types import SimpleNamespace
        from .spectral_conv import SpectralConv
        kept = [int(f.shape[0]) for f in factors[2:]]
        t3 = SpectralConv._tucker_core_times_modes(SimpleNamespace(core=core, factors=list(factors)), kept)
        m = int(t3.shape[2])
        w1 = mode_gemm(factors[0], t3, m)
        w = mode_gemm(w1, factors[1].transpose(0, 1), m)
        return w.reshape(w.shape[0], w.shape[1], *kept)