"""The rest of an FNO block on the engine: what FNOBlocks.forward_with_postactivation
(/root/reference/neuralop/layers/fno_block.py:377-414) does around the spectral convolution.

    x_skip_fno = fno_skips[i](x)                      1 x 1 linear skip (skip_connections.py:119-169) -> fused_linear
    x          = gelu(convs[i](x) + x_skip_fno)       -> SpectralConv.forward_fused: add + GELU in the inverse
                                                         transform's store path (sc_layer_forward_ex)
    x          = channel_mlp[i](x) + gate * x_in      -> fused_channel_mlp: both 1 x 1 convolutions, the GELU between
    x          = gelu(x)            (not the last block)   them, the soft-gating skip and the closing GELU in ONE pass
                                                         over the tensor (sc_pointwise_mlp_forward / _backward)

``fused_block_forward(blocks, x, index)`` runs exactly that (three engine passes forward) on the parameters of an ``FNOBlocks``-shaped module (the
verbatim reference class, built with ``conv_module=neuraloperator_amd.SpectralConv``): same result as
``blocks(x, index)``; configurations outside its scope (normalisation layers, pre-activation, tanh stabiliser, a
resolution change, other skip types) take the module's own forward.  Session 2: pre-activation blocks and blocks with
normalisation layers run the same engine passes composed by autograd (``_fused_block_variant``)."""
import os

import torch
import torch.nn.functional as F

from . import _lib
from .engine import _require_gpu, _stream


class PointwiseMLPFn(torch.autograd.Function):
    """out = act(W2 gelu(W1 x + b1) + b2 + gate * skip_src) and all of its gradients, one pass each way
    (sc_kernels_pmlp.h); saves only its inputs."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, skip_src, gate, act):
        _require_gpu(x, "x")
        shape = x.shape
        b, ci = int(shape[0]), int(shape[1])
        s = 1
        for v in shape[2:]:
            s *= int(v)
        ch, co = int(w1.shape[0]), int(w2.shape[0])
        xc = x.contiguous()
        w1c, w2c = w1.reshape(ch, ci).contiguous(), w2.reshape(co, ch).contiguous()
        b1c = None if b1 is None else b1.contiguous()
        b2c = None if b2 is None else b2.contiguous()
        skc = None if skip_src is None else skip_src.contiguous()
        gtc = None if gate is None else gate.reshape(co).contiguous()
        out = torch.empty((b, co, *shape[2:]), dtype=torch.float32, device=x.device)
        p = lambda t: 0 if t is None else t.data_ptr()
        with torch.cuda.device(x.device):
            _lib.get_lib().pointwise_mlp_forward(b, ci, ch, co, s, act, p(xc), p(w1c), p(b1c), p(w2c), p(b2c), p(skc),
                                                 p(gtc), p(out), _stream())
        ctx.save_for_backward(xc, w1c, b1c, w2c, b2c, skc, gtc)
        ctx.cfg = (b, ci, ch, co, s, act, tuple(w1.shape), tuple(w2.shape), None if gate is None else tuple(gate.shape))
        return out

    @staticmethod
    def backward(ctx, gout):
        xc, w1c, b1c, w2c, b2c, skc, gtc = ctx.saved_tensors
        b, ci, ch, co, s, act, w1_shape, w2_shape, gate_shape = ctx.cfg
        lib = _lib.get_lib()
        dev = xc.device
        gout = gout.contiguous()
        gx = torch.empty_like(xc)
        gw1, gw2 = torch.empty_like(w1c), torch.empty_like(w2c)
        gb1 = None if b1c is None else torch.empty_like(b1c)
        gb2 = None if b2c is None else torch.empty_like(b2c)
        gsk = None if skc is None else torch.empty_like(skc)
        ggt = None if gtc is None else torch.empty_like(gtc)
        ws = torch.empty(lib.pointwise_mlp_workspace_bytes(b, ci, ch, co, s, act), dtype=torch.uint8, device=dev)
        p = lambda t: 0 if t is None else t.data_ptr()
        with torch.cuda.device(dev):
            lib.pointwise_mlp_backward(b, ci, ch, co, s, act, p(xc), p(w1c), p(b1c), p(w2c), p(b2c), p(skc), p(gtc),
                                       p(gout), p(gx), p(gw1), p(gb1), p(gw2), p(gb2), p(gsk), p(ggt), p(ws), _stream())
        return (gx, gw1.reshape(w1_shape), gb1, gw2.reshape(w2_shape), gb2, gsk,
                None if ggt is None else ggt.reshape(gate_shape), None)


class PointwiseLinearFn(torch.autograd.Function):
    """out = conv1x1(x, w, bias) (the block's linear skip, skip_connections.py:119-169) and its gradients, one pass
    each way (k_plin_fwd / k_plin_bwd)."""

    @staticmethod
    def forward(ctx, x, w, bias):
        _require_gpu(x, "x")
        shape = x.shape
        b, ci, co = int(shape[0]), int(shape[1]), int(w.shape[0])
        s = 1
        for v in shape[2:]:
            s *= int(v)
        xc, wc = x.contiguous(), w.reshape(co, ci).contiguous()
        bc = None if bias is None else bias.contiguous()
        out = torch.empty((b, co, *shape[2:]), dtype=torch.float32, device=x.device)
        with torch.cuda.device(x.device):
            _lib.get_lib().pointwise_linear_forward(b, ci, co, s, xc.data_ptr(), wc.data_ptr(),
                                                    0 if bc is None else bc.data_ptr(), out.data_ptr(), _stream())
        ctx.save_for_backward(xc, wc)
        ctx.cfg = (b, ci, co, s, tuple(w.shape), bias is not None)
        return out

    @staticmethod
    def backward(ctx, gout):
        xc, wc = ctx.saved_tensors
        b, ci, co, s, w_shape, has_bias = ctx.cfg
        lib = _lib.get_lib()
        gout = gout.contiguous()
        gx, gw = torch.empty_like(xc), torch.empty_like(wc)
        gb = torch.empty(co, dtype=torch.float32, device=xc.device) if has_bias else None
        ws = torch.empty(lib.pointwise_linear_workspace_bytes(b, ci, co, s), dtype=torch.uint8, device=xc.device)
        with torch.cuda.device(xc.device):
            lib.pointwise_linear_backward(b, ci, co, s, xc.data_ptr(), wc.data_ptr(), gout.data_ptr(), gx.data_ptr(),
                                          gw.data_ptr(), 0 if gb is None else gb.data_ptr(), ws.data_ptr(), _stream())
        return gx, gw.reshape(w_shape), gb


_PLX_CH = (32, 64, 128)                                   # channel counts of the extended passes (sc_kernels_plinx.h)


def _pw_size(s):
    """points per (sample, channel) image the pointwise kernels take: 32-pixel tiles, lane offsets as 32-bit byte counts
    (csrc/sc_engine.cpp SC_PW_MAX_SPATIAL); anything else keeps the ATen route"""
    return s % 32 == 0 and s < (1 << 28)


def _plx_ok(*cs):
    return all(c in _PLX_CH for c in cs)


class PointwiseLinearXFn(torch.autograd.Function):
    """out = conv1x1(x, w, bias) for any channel counts in {32, 64, 128} (round 6: sc_pointwise_linear_forward_ex /
    _backward_ex -- rectangular maps and the 128-channel gradient the square kernels of PointwiseLinearFn lack)."""

    @staticmethod
    def forward(ctx, x, w, bias):
        _require_gpu(x, "x")
        shape = x.shape
        b, ci, co = int(shape[0]), int(shape[1]), int(w.shape[0])
        s = x[0, 0].numel()
        xc, wc = x.contiguous(), w.reshape(co, ci).contiguous()
        bc = None if bias is None else bias.contiguous()
        out = torch.empty((b, co, *shape[2:]), dtype=torch.float32, device=x.device)
        with torch.cuda.device(x.device):
            _lib.get_lib().pointwise_linear_forward_ex(b, ci, co, s, 0, xc.data_ptr(), wc.data_ptr(),
                                                       0 if bc is None else bc.data_ptr(), 0, 0, out.data_ptr(), 0, _stream())
        ctx.save_for_backward(xc, wc)
        ctx.cfg = (b, ci, co, s, tuple(w.shape), bias is not None)
        return out

    @staticmethod
    def backward(ctx, gout):
        xc, wc = ctx.saved_tensors
        b, ci, co, s, w_shape, has_bias = ctx.cfg
        lib = _lib.get_lib()
        gout = gout.contiguous()
        gx, gw = torch.empty_like(xc), torch.empty_like(wc)
        gb = torch.empty(co, dtype=torch.float32, device=xc.device) if has_bias else None
        ws = torch.empty(lib.pointwise_linear_workspace_bytes_ex(b, ci, co, s), dtype=torch.uint8, device=xc.device)
        with torch.cuda.device(xc.device):
            lib.pointwise_linear_backward_ex(b, ci, co, s, 0, xc.data_ptr(), wc.data_ptr(), gout.data_ptr(), 0, 0, 0, 0, 0,
                                             gx.data_ptr(), gw.data_ptr(), 0 if gb is None else gb.data_ptr(), 0, 0,
                                             ws.data_ptr(), _stream())
        return gx, gw.reshape(w_shape), gb


class PointwiseMLP2Fn(torch.autograd.Function):
    """The ChannelMLP pass of PointwiseMLPFn as TWO engine passes each way (round 6): channel counts without a one-pass
    kernel -- hidden 128, configs[4]'s width.  The hidden activations cross memory once, as their pre-activation; the
    GELUs, the soft-gating skip and their derivatives ride in the passes (csrc/sc_kernels_plinx.h):

        h_pre = W1 x + b1                                  out = act(W2 gelu(h_pre) + b2 + gate (.) skip_src)
        ghp   = (W2^T g) (.) gelu'(h_pre), g = gout (.) act'(z_pre);  gx = W1^T ghp;  weight / bias / gate gradients in the
        same two passes."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, skip_src, gate, act):
        _require_gpu(x, "x")
        lib = _lib.get_lib()
        shape = x.shape
        b, ci = int(shape[0]), int(shape[1])
        s = x[0, 0].numel()
        ch, co = int(w1.shape[0]), int(w2.shape[0])
        xc = x.contiguous()
        w1c, w2c = w1.reshape(ch, ci).contiguous(), w2.reshape(co, ch).contiguous()
        b1c = None if b1 is None else b1.contiguous()
        b2c = None if b2 is None else b2.contiguous()
        skc = None if skip_src is None else skip_src.contiguous()
        gtc = None if gate is None else gate.reshape(co).contiguous()
        dev = x.device
        hpre = torch.empty((b, ch, *shape[2:]), dtype=torch.float32, device=dev)
        out = torch.empty((b, co, *shape[2:]), dtype=torch.float32, device=dev)
        zpre = torch.empty_like(out) if act == _lib.SC_ACT_GELU else None
        p = lambda t: 0 if t is None else t.data_ptr()
        with torch.cuda.device(dev):
            st = _stream()
            lib.pointwise_linear_forward_ex(b, ci, ch, s, 0, p(xc), p(w1c), p(b1c), 0, 0, p(hpre), 0, st)
            fl = _lib.SC_PLX_XACT | (_lib.SC_PLX_ACT if zpre is not None else 0)
            lib.pointwise_linear_forward_ex(b, ch, co, s, fl, p(hpre), p(w2c), p(b2c), p(skc), p(gtc), p(out), p(zpre), st)
        ctx.save_for_backward(xc, w1c, w2c, skc, gtc, hpre, zpre)
        ctx.cfg = (b, ci, ch, co, s, tuple(w1.shape), tuple(w2.shape), None if gate is None else tuple(gate.shape),
                   b1 is not None, b2 is not None)
        return out

    @staticmethod
    def backward(ctx, gout):
        xc, w1c, w2c, skc, gtc, hpre, zpre = ctx.saved_tensors
        b, ci, ch, co, s, w1_shape, w2_shape, gate_shape, has_b1, has_b2 = ctx.cfg
        lib = _lib.get_lib()
        dev = xc.device
        gout = gout.contiguous()
        new = lambda n: torch.empty(n, dtype=torch.float32, device=dev)
        ghp, gx = torch.empty_like(hpre), torch.empty_like(xc)
        gw1, gw2 = torch.empty_like(w1c), torch.empty_like(w2c)
        gb1, gb2 = (new(ch) if has_b1 else None), (new(co) if has_b2 else None)
        gsk = None if skc is None else torch.empty_like(skc)
        ggt = None if gtc is None else torch.empty_like(gtc)
        ws = torch.empty(max(lib.pointwise_linear_workspace_bytes_ex(b, ch, co, s), lib.pointwise_linear_workspace_bytes_ex(b, ci, ch, s)),
                         dtype=torch.uint8, device=dev)
        p = lambda t: 0 if t is None else t.data_ptr()
        with torch.cuda.device(dev):
            st = _stream()
            fl = _lib.SC_PLX_XACT | _lib.SC_PLX_XGRAD | (_lib.SC_PLX_PRO if zpre is not None else 0)
            lib.pointwise_linear_backward_ex(b, ch, co, s, fl, p(hpre), p(w2c), p(gout), p(zpre), p(hpre), p(skc), p(gtc), 0,
                                             p(ghp), p(gw2), p(gb2), p(gsk), p(ggt), p(ws), st)
            lib.pointwise_linear_backward_ex(b, ci, ch, s, 0, p(xc), p(w1c), p(ghp), 0, 0, 0, 0, 0, p(gx), p(gw1), p(gb1), 0, 0,
                                             p(ws), st)
        return (gx, gw1.reshape(w1_shape), gb1, gw2.reshape(w2_shape), gb2, gsk,
                None if ggt is None else ggt.reshape(gate_shape), None)


def fused_linear(x, w, bias=None):
    """1 x 1 convolution over the channels on the engine for 32 / 64 / 128 input and output channels (any pair, with
    gradients: round 6) and a pixel count that is a multiple of 32, ``F.conv1d`` otherwise."""
    ci, co = int(x.shape[1]), int(w.shape[0])
    s = x[0, 0].numel()
    needs_grad = torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in (x, w, bias))
    ok = (32, 64) if needs_grad else (32, 64, 128)
    if _on_engine(x) and x.dtype == torch.float32 and ci == co and ci in ok and _pw_size(s):
        return PointwiseLinearFn.apply(x, w, bias)
    if _on_engine(x) and x.dtype == torch.float32 and _plx_ok(ci, co) and _pw_size(s):       # round 6: any pair of 32 / 64 / 128
        return PointwiseLinearXFn.apply(x, w, bias)
    shape = x.shape
    return F.conv1d(x.reshape(shape[0], ci, -1), w.reshape(co, ci, 1), bias).reshape(shape[0], co, *shape[2:])


_SHAPES = {(32, 32, 32), (64, 32, 64), (64, 64, 64), (128, 64, 128)}
_SHAPES_BWD = {(32, 32, 32), (64, 32, 64), (64, 64, 64)}      # (128, 64, 128): forward kernel only


def _on_engine(t):
    return t.is_cuda


def fused_channel_mlp(x, w1, b1, w2, b2, skip_src=None, gate=None, activation=None):
    """``act(conv1x1(gelu(conv1x1(x, w1, b1)), w2, b2) + gate * skip_src)``: w1 / w2 are Conv1d weights
    (out, in[, 1]); gate a per-channel weight of any broadcastable shape with ``out`` elements; activation None or
    "gelu".  One engine pass when the channel counts have a one-pass kernel and the pixel count is a multiple of 32, two
    engine passes each way for any other channel counts in {32, 64, 128} (round 6), the plain composition of the same
    operations otherwise."""
    ci, ch, co = int(x.shape[1]), int(w1.shape[0]), int(w2.shape[0])
    s = x[0, 0].numel()
    needs_grad = torch.is_grad_enabled() and any(t is not None and t.requires_grad
                                                   for t in (x, w1, b1, w2, b2, skip_src, gate))
    fits = _on_engine(x) and x.dtype == torch.float32 and _pw_size(s) and (skip_src is None) == (gate is None) and \
        (ci, ch, co) in (_SHAPES_BWD if needs_grad else _SHAPES)
    if fits:
        act = _lib.SC_ACT_GELU if activation == "gelu" else _lib.SC_ACT_NONE
        return PointwiseMLPFn.apply(x, w1, b1, w2, b2, skip_src, gate, act)
    if _on_engine(x) and x.dtype == torch.float32 and _pw_size(s) and (skip_src is None) == (gate is None) and \
            _plx_ok(ci, ch, co):                         # round 6: two engine passes each way (hidden 128, 128 channels)
        act = _lib.SC_ACT_GELU if activation == "gelu" else _lib.SC_ACT_NONE
        return PointwiseMLP2Fn.apply(x, w1, b1, w2, b2, skip_src, gate, act)
    shape = x.shape
    h = F.gelu(F.conv1d(x.reshape(shape[0], ci, -1), w1.reshape(ch, ci, 1), b1))
    z = F.conv1d(h, w2.reshape(co, ch, 1), b2).reshape(shape[0], co, *shape[2:])
    if skip_src is not None:
        z = z + gate.reshape(1, co, *(1,) * (x.ndim - 2)) * skip_src
    return F.gelu(z) if activation == "gelu" else z


_PBLOCK_SHAPES = {(32, 32), (64, 32), (64, 64)}          # (channels, hidden): sc_pointwise_block_forward's kernels
_NO_PBLOCK = os.environ.get("SC_BLOCK_NO_PBLOCK") == "1"   # A-B: the three-pass forward of round 2


class FusedBlockFn(torch.autograd.Function):
    """One default FNO block as ONE autograd node: linear skip -> Fourier layer with the add + GELU epilogue ->
    pointwise MLP pass.  The block input feeds three branches; as separate nodes autograd adds their three gradients
    with two tensor-sized passes (6 tensor reads / writes).  Here the backward chains them through the store paths:
    the MLP pass' skip gradient is the addend of the linear skip's backward, whose result is the addend of the last
    transform of the spectral convolution's backward (sc_pointwise_linear_backward / sc_layer_backward_ex)."""

    @staticmethod
    def forward(ctx, x, cw, cb, lw, lb, w1, b1, w2, b2, gate, last, n_modes_attr, max_n_modes_attr, fft_norm, flags):
        from . import engine
        from .modes import kept_block
        _require_gpu(x, "x")
        lib = _lib.get_lib()
        dev = x.device
        x = x.contiguous().float()
        b, c = int(x.shape[0]), int(x.shape[1])
        spatial = [int(v) for v in x.shape[2:]]
        s = 1
        for v in spatial:
            s *= v
        ch = int(w1.shape[0])
        act = _lib.SC_ACT_NONE if last else _lib.SC_ACT_GELU
        p = lambda t: 0 if t is None else t.data_ptr()
        lwc, w1c, w2c = lw.detach().reshape(c, c).contiguous(), w1.detach().reshape(ch, c).contiguous(), w2.detach().reshape(c, ch).contiguous()
        lbc = None if lb is None else lb.detach().contiguous()
        b1c, b2c = (None if t is None else t.detach().contiguous() for t in (b1, b2))
        gtc = gate.detach().reshape(c).contiguous()
        cwc = cw.detach().to(torch.complex64).contiguous()
        cbf = None if cb is None else cb.detach().reshape(-1).float().contiguous()
        kept, w_start = kept_block(spatial, n_modes_attr, max_n_modes_attr)
        plan = engine.get_plan(dev, spatial, kept, fft_norm, flags)
        L = lib.layer_desc(b, c, c, list(cwc.shape[2:]), w_start)
        with torch.cuda.device(dev):
            st = _stream()
            ws = engine._ws(lib.layer_workspace_bytes(plan, L), dev)
            y = torch.empty_like(x)
            pre = None if last else torch.empty_like(x)
            xhat = torch.empty((b, c, *kept, 2), dtype=torch.float32, device=dev)
            out = torch.empty_like(x)
            hpre = zpre = None
            if (c, ch, c) not in _SHAPES_BWD:
                # round 6: channel counts without a one-pass kernel (hidden 128, 128 channels): the same block as engine
                # passes of csrc/sc_kernels_plinx.h -- skip, Fourier layer with the add + GELU in its store path, fc1, fc2
                # with the GELUs and the soft-gating skip in their load / store paths; nothing elementwise in between
                # the linear skip and the Fourier layer's add + GELU as ONE pass: y = act(W_s x + b_s + 1 (.) conv) through the
                # map's gated-skip path (gate = ones, skip = the plain spectral convolution) -- the skip tensor is never
                # written and, where the transform has no fused epilogue (two-pass routes: configs[4]'s 1024^2), the
                # streaming k_epilogue pass goes too: 6 tensor-sized reads / writes -> 4
                conv = torch.empty_like(x)
                lib.layer_forward_ex(plan, L, p(x), torch.view_as_real(cwc).data_ptr(), p(cbf), 0, 0, _lib.SC_ACT_NONE, p(conv),
                                     p(xhat), p(ws), st)
                ones = torch.ones(c, dtype=torch.float32, device=dev)
                lib.pointwise_linear_forward_ex(b, c, c, s, 0 if last else _lib.SC_PLX_ACT, p(x), p(lwc), p(lbc), p(conv), p(ones),
                                                p(y), p(pre), st)
                del conv
                hpre = torch.empty((b, ch, *spatial), dtype=torch.float32, device=dev)
                zpre = None if last else torch.empty_like(x)
                lib.pointwise_linear_forward_ex(b, c, ch, s, 0, p(y), p(w1c), p(b1c), 0, 0, p(hpre), 0, st)
                lib.pointwise_linear_forward_ex(b, ch, c, s, _lib.SC_PLX_XACT | (0 if last else _lib.SC_PLX_ACT), p(hpre), p(w2c),
                                                p(b2c), p(x), p(gtc), p(out), p(zpre), st)
            elif (c, ch) in _PBLOCK_SHAPES and not _NO_PBLOCK:
                # session 2: a plain inverse transform, then ONE pointwise pass for skip + add + GELU + MLP + gate (the
                # skip is never written, y is not read back: 1 + 5 tensor-sized passes instead of 8); y and pre are
                # the same tensors as before, so the backward below does not change
                conv = torch.empty_like(x)
                lib.layer_forward_ex(plan, L, p(x), torch.view_as_real(cwc).data_ptr(), p(cbf), 0, 0, _lib.SC_ACT_NONE, p(conv),
                                     p(xhat), p(ws), st)
                # round 6: `pre` receives gelu'(conv + skip) (SC_ACT_GELU_DGRAD: one evaluation of the transcendentals gives
                # the activation and its derivative), the backward pass multiplies by it instead of evaluating gelu' again
                if not last:
                    act = _lib.SC_ACT_GELU_DGRAD
                lib.pointwise_block_forward(b, c, ch, s, act, p(conv), p(x), p(lwc), p(lbc), p(w1c), p(b1c), p(w2c), p(b2c),
                                            p(gtc), p(y), p(pre), p(out), st)
            else:
                skip = torch.empty_like(x)
                lib.pointwise_linear_forward(b, c, c, s, p(x), p(lwc), p(lbc), p(skip), st)
                lib.layer_forward_ex(plan, L, p(x), torch.view_as_real(cwc).data_ptr(), p(cbf), p(skip), p(pre), act, p(y),
                                     p(xhat), p(ws), st)
                lib.pointwise_mlp_forward(b, c, ch, c, s, act, p(y), p(w1c), p(b1c), p(w2c), p(b2c), p(x), p(gtc), p(out), st)
        ctx.save_for_backward(x, y, pre, xhat, cwc, lwc, w1c, b1c, w2c, b2c, gtc, hpre, zpre)
        ctx.cfg = (plan, L, b, c, ch, s, act, tuple(cw.shape), None if cb is None else tuple(cb.shape), tuple(lw.shape),
                   lb is not None, tuple(w1.shape), tuple(w2.shape), tuple(gate.shape))
        return out

    @staticmethod
    def backward(ctx, gout):
        from . import engine
        x, y, pre, xhat, cwc, lwc, w1c, b1c, w2c, b2c, gtc, hpre, zpre = ctx.saved_tensors
        plan, L, b, c, ch, s, act, cw_shape, cb_shape, lw_shape, has_lb, w1_shape, w2_shape, gate_shape = ctx.cfg
        lib = _lib.get_lib()
        dev = x.device
        p = lambda t: 0 if t is None else t.data_ptr()
        gout = gout.contiguous().float()
        with torch.cuda.device(dev):
            st = _stream()
            # pointwise MLP pass: gy, and the soft-gating branch's gradient of the block input (acc)
            gy = torch.empty_like(x)
            fused_bwd = hpre is None and lib.pointwise_block_backward_supported(b, c, ch, s)
            acc = None if fused_bwd else torch.empty_like(x)
            gw1, gw2 = torch.empty_like(w1c), torch.empty_like(w2c)
            gb1 = None if b1c is None else torch.empty_like(b1c)
            gb2 = None if b2c is None else torch.empty_like(b2c)
            ggt = torch.empty_like(gtc)
            acc2, glw = torch.empty_like(x), torch.empty_like(lwc)
            glb = torch.empty(c, dtype=torch.float32, device=dev) if has_lb else None
            if hpre is not None:
                # round 6, the two-pass form: fc2 (gradient through the closing GELU, the soft-gating branch -> acc, ghp =
                # the gradient of the hidden PRE-activation), fc1 (gz = the gradient through the Fourier layer's GELU),
                # the linear skip with acc as its addend -- three calls of sc_pointwise_linear_backward_ex
                ghp = torch.empty_like(hpre)
                wsx = torch.empty(max(lib.pointwise_linear_workspace_bytes_ex(b, ch, c, s),
                                      lib.pointwise_linear_workspace_bytes_ex(b, c, ch, s),
                                      lib.pointwise_linear_workspace_bytes_ex(b, c, c, s)), dtype=torch.uint8, device=dev)
                fl2 = _lib.SC_PLX_XACT | _lib.SC_PLX_XGRAD | (_lib.SC_PLX_PRO if zpre is not None else 0)
                lib.pointwise_linear_backward_ex(b, ch, c, s, fl2, p(hpre), p(w2c), p(gout), p(zpre), p(hpre), p(x), p(gtc), 0,
                                                 p(ghp), p(gw2), p(gb2), p(acc), p(ggt), p(wsx), st)
                lib.pointwise_linear_backward_ex(b, c, ch, s, _lib.SC_PLX_XGRAD if pre is not None else 0, p(y), p(w1c), p(ghp),
                                                 0, p(pre), 0, 0, 0, p(gy), p(gw1), p(gb1), 0, 0, p(wsx), st)
                gz = gy
                lib.pointwise_linear_backward_ex(b, c, c, s, 0, p(x), p(lwc), p(gz), 0, 0, 0, 0, p(acc), p(acc2), p(glw), p(glb),
                                                 0, 0, p(wsx), st)
            elif fused_bwd:
                # round 6: the data path of the linear skip rides in the MLP pass (acc2 = W_s^T gz + the soft-gating
                # branch's gradient never crosses memory as two tensors); the skip's weight gradient is a pass of its own
                # over gz and x (sc_pointwise_linear_backward_ex without gx): 10 tensor-sized reads / writes -> 7
                ws = torch.empty(lib.pointwise_mlp_workspace_bytes(b, c, ch, c, s, act), dtype=torch.uint8, device=dev)
                lib.pointwise_block_backward(b, c, ch, s, act, p(y), p(pre), p(x), p(lwc), p(w1c), p(b1c), p(w2c), p(b2c), p(gtc),
                                             p(gout), p(gy), p(acc2), p(gw1), p(gb1), p(gw2), p(gb2), p(ggt), p(ws), st)
                gz = gy
                wsl = torch.empty(lib.pointwise_linear_workspace_bytes_ex(b, c, c, s), dtype=torch.uint8, device=dev)
                lib.pointwise_linear_backward_ex(b, c, c, s, 0, p(x), p(lwc), p(gz), 0, 0, 0, 0, 0, 0, p(glw), p(glb), 0, 0,
                                                 p(wsl), st)
            else:
                ws = torch.empty(lib.pointwise_mlp_workspace_bytes(b, c, ch, c, s, act), dtype=torch.uint8, device=dev)
                # (with the Fourier layer's pre-activation the pass returns the gradient THROUGH its GELU: gy is gz)
                lib.pointwise_mlp_backward(b, c, ch, c, s, act, p(y), p(w1c), p(b1c), p(w2c), p(b2c), p(x), p(gtc), p(gout),
                                           p(gy), p(gw1), p(gb1), p(gw2), p(gb2), p(acc), p(ggt), p(ws), st, x_pre=p(pre))
                gz = gy
                # linear skip: W^T gz + acc
                ws2 = torch.empty(lib.pointwise_linear_workspace_bytes(b, c, c, s), dtype=torch.uint8, device=dev)
                lib.pointwise_linear_backward(b, c, c, s, p(x), p(lwc), p(gz), p(acc2), p(glw), p(glb), p(ws2), st, addend=p(acc))
            # spectral convolution: its input gradient + acc2 in the store path of the last transform
            gx = torch.empty_like(x)
            full = all(L.w_start[d] == 0 for d in range(len(cw_shape) - 2)) and tuple(xhat.shape[2:-1]) == tuple(cw_shape[2:])
            gcw = (torch.empty if full else torch.zeros)((*cw_shape, 2), dtype=torch.float32, device=dev)
            gcb = None if cb_shape is None else torch.empty(c, dtype=torch.float32, device=dev)
            ws3 = engine._ws(lib.layer_workspace_bytes(plan, L), dev)
            lib.layer_backward_ex(plan, L, p(gz), p(xhat), torch.view_as_real(cwc).data_ptr(), p(gx), p(gcw), p(gcb), p(acc2),
                                  p(ws3), st)
        return (gx, torch.view_as_complex(gcw), None if gcb is None else gcb.reshape(cb_shape), glw.reshape(lw_shape), glb,
                gw1.reshape(w1_shape), gb1, gw2.reshape(w2_shape), gb2, ggt.reshape(gate_shape),
                None, None, None, None, None)


def _block_in_scope(blocks, index, output_shape, variants=False):
    """the default block (post-activation, no normalisation layers); ``variants``: also pre-activation and / or
    normalisation layers (session 2: the same engine passes, composed -- fused_block_forward)"""
    if output_shape is not None:
        return False
    if not variants and (getattr(blocks, "preactivation", False) or getattr(blocks, "norm", None) is not None):
        return False
    if getattr(blocks, "stabilizer", None) is not None or getattr(blocks, "complex_data", False):
        return False
    conv = blocks.convs[index]
    if not hasattr(conv, "forward_fused") or getattr(conv, "resolution_scaling_factor", None) is not None:
        return False
    if not getattr(blocks, "use_channel_mlp", False) or blocks.fno_skips is None or blocks.channel_mlp_skips is None:
        return False
    mlp, gskip, fskip = blocks.channel_mlp[index], blocks.channel_mlp_skips[index], blocks.fno_skips[index]
    if len(getattr(mlp, "fcs", ())) != 2 or getattr(mlp, "dropout", None) is not None:
        return False
    if type(gskip).__name__ != "SoftGating" or getattr(gskip, "bias", None) is not None:
        return False
    if type(fskip).__name__ != "Flattened1dConv":
        return False
    act = getattr(blocks, "non_linearity", None)
    return act is F.gelu and getattr(mlp, "non_linearity", None) is F.gelu


def _fused_block_variant(blocks, x, index, last, conv, fc1, fc2, lin, pre, norm):
    """Pre-activation (fno_block.py:416-458) and / or normalisation layers (:399-403, :408-409, :421-422, :447-448) on
    the engine passes of the default block, composed with autograd between them:

      pre-activation, no norm   x = gelu(x) [one ATen pass]; 1 x 1 skip; Fourier layer with the add (+ GELU unless last)
                                in its store path; pointwise MLP pass with the soft-gating skip of the ACTIVATED x, no
                                closing activation
      normalisation layers      they sit between the convolution and the add / behind the MLP, so the add and the
                                activations around them stay ATen calls; the 1 x 1 skip, the spectral convolution and
                                the pointwise MLP pass (both 1 x 1 convolutions, the GELU between them, the soft-gating
                                skip) run on the engine

    Same operations in the same order as the reference's two forward methods."""
    act = blocks.non_linearity
    nn_ = getattr(blocks, "n_norms", 2)
    gate = blocks.channel_mlp_skips[index].weight
    if pre:
        x = act(x)
        if norm is not None:
            x = norm[nn_ * index](x)
    x_skip_fno = fused_linear(x, lin.weight, lin.bias)
    if norm is None:                                     # pre-activation only: the add (+ activation) in the store path
        y = conv.forward_fused(x, x_skip_fno, activation=None if last else "gelu")
        return fused_channel_mlp(y, fc1.weight, fc1.bias, fc2.weight, fc2.bias, skip_src=x, gate=gate, activation=None)
    x_fno = conv(x)
    if not pre:
        x_fno = norm[nn_ * index](x_fno)
    y = x_fno + x_skip_fno
    if not last:
        y = act(y)
    if pre:
        y = norm[nn_ * index + 1](y)
    out = fused_channel_mlp(y, fc1.weight, fc1.bias, fc2.weight, fc2.bias, skip_src=x, gate=gate, activation=None)
    if not pre:
        out = norm[nn_ * index + 1](out)
        if not last:
            out = act(out)
    return out


def fused_block_forward(blocks, x, index=0, output_shape=None):
    """FNOBlocks.forward_with_postactivation (fno_block.py:377-414) for block ``index`` in two engine passes + the 1 x 1
    skip convolution (a plain library GEMM); the module's own forward outside the scope described in the module
    docstring."""
    if not _block_in_scope(blocks, index, output_shape, variants=True):
        return blocks(x, index, output_shape=output_shape)
    last = index >= blocks.n_layers - 1
    conv = blocks.convs[index]
    fc1, fc2 = blocks.channel_mlp[index].fcs
    lin = blocks.fno_skips[index].conv
    pre, norm = bool(getattr(blocks, "preactivation", False)), getattr(blocks, "norm", None)
    if pre or norm is not None:
        return _fused_block_variant(blocks, x, index, last, conv, fc1, fc2, lin, pre, norm)
    c, ch = int(x.shape[1]), int(fc1.weight.shape[0])
    from .factorized import DenseWeight
    one_node = _on_engine(x) and x.dtype == torch.float32 and _pw_size(x[0, 0].numel()) and \
        (((c, ch, c) in _SHAPES_BWD and c in (32, 64)) or _plx_ok(c, ch)) and int(lin.weight.shape[0]) == c and \
        isinstance(conv.weight, DenseWeight) and not conv.separable and conv.fno_block_precision == "full" and \
        conv.in_channels == conv.out_channels == c
    if one_node:
        return FusedBlockFn.apply(x, conv.weight.tensor, conv.bias, lin.weight, lin.bias, fc1.weight, fc1.bias, fc2.weight,
                                  fc2.bias, blocks.channel_mlp_skips[index].weight, last, list(conv.n_modes),
                                  list(conv.max_n_modes), conv.fft_norm, conv.engine_flags)                                        # 1 x 1 convolution, no bias by default
    x_skip_fno = fused_linear(x, lin.weight, lin.bias)
    y = conv.forward_fused(x, x_skip_fno, activation=None if last else "gelu")
    return fused_channel_mlp(y, fc1.weight, fc1.bias, fc2.weight, fc2.bias, skip_src=x,
                             gate=blocks.channel_mlp_skips[index].weight, activation=None if last else "gelu")
