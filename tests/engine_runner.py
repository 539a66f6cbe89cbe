"""Drive the engine through its C-ABI with torch tensors on any device.

Used by the CPU tier with the host-emulation build (tests/emu, CPU tensors) and by the
``-m gpu`` tier with the real libsc_engine.so (CUDA tensors).  Pointer level only -- this
is the same call sequence neuraloperator_amd.spectral_conv issues.
"""
import os
import subprocess
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from neuraloperator_amd import _lib  # noqa: E402
from neuraloperator_amd.modes import kept_block  # noqa: E402

EMU_DIR = os.path.join(ROOT, "tests", "emu")
EMU_LIB = os.path.join(EMU_DIR, "libsc_engine_emu.so")


def emu_lib():
    """Build (if stale) and load the host-emulation library.  TEST ONLY.
    SC_EMU_LIB=<path> loads a prebuilt variant instead (kernel shape A-B builds: g++ ... -DSC_G8_VARIANT=n)."""
    if os.environ.get("SC_EMU_LIB"):
        return _lib.ScEngineLib(os.environ["SC_EMU_LIB"])
    srcs = [os.path.join(ROOT, "neuraloperator_amd", "csrc", f)
            for f in os.listdir(os.path.join(ROOT, "neuraloperator_amd", "csrc"))
            if f.endswith((".h", ".cpp"))]
    srcs += [os.path.join(EMU_DIR, "sc_emu_runtime.cpp"), os.path.join(ROOT, "include", "sc_engine.h")]
    import fcntl
    with open(os.path.join(EMU_DIR, ".build.lock"), "w") as lock:      # pytest-xdist workers: one builds, the rest wait
        fcntl.flock(lock, fcntl.LOCK_EX)
        stale = (not os.path.isfile(EMU_LIB)) or any(
            os.path.getmtime(s) > os.path.getmtime(EMU_LIB) for s in srcs)
        if stale:
            subprocess.check_call(["sh", os.path.join(EMU_DIR, "build_emu.sh")], stdout=subprocess.DEVNULL)
    return _lib.ScEngineLib(EMU_LIB)


def _stream(dev):
    return torch.cuda.current_stream().cuda_stream if dev.type == "cuda" else 0


def _ptr(t):
    return 0 if t is None else t.data_ptr()


def layer_fwd_bwd(lib, x, w, bias, g, n_modes_attr, max_n_modes_attr, flags=0, fft_norm="forward"):
    """One dense SpectralConv forward+backward through sc_layer_forward/backward.
    Returns (y, gx, gw, gbias, xhat) as tensors on x's device."""
    dev = x.device
    b, cin = x.shape[:2]
    cout = w.shape[1]
    spatial = list(x.shape[2:])
    kept, w_start = kept_block(spatial, n_modes_attr, max_n_modes_attr)
    plan = lib.plan_create(spatial, kept, fft_norm=fft_norm, flags=flags)
    try:
        L = lib.layer_desc(b, cin, cout, list(w.shape[2:]), w_start)
        ws = torch.empty(lib.layer_workspace_bytes(plan, L), dtype=torch.uint8, device=dev)
        x = x.contiguous()
        w = w.contiguous()
        g = g.contiguous()
        wv = torch.view_as_real(w)
        # SC_PLAN_IO_BF16: x and g are bfloat16 tensors, y and gx come back as bfloat16
        y = torch.empty(b, cout, *spatial, dtype=x.dtype, device=dev)
        xhat = torch.empty(b, cin, *kept, 2, dtype=torch.float32, device=dev)
        bias_flat = None if bias is None else bias.reshape(-1).contiguous()
        st = _stream(dev)
        lib.layer_forward(plan, L, _ptr(x), _ptr(wv), _ptr(bias_flat), _ptr(y), _ptr(xhat), _ptr(ws), st)
        gx = torch.empty_like(x)
        gw = torch.zeros_like(wv)
        gb = torch.empty(cout, dtype=torch.float32, device=dev)
        lib.layer_backward(plan, L, _ptr(g), _ptr(xhat), _ptr(wv), _ptr(gx), _ptr(gw), _ptr(gb), _ptr(ws), st)
        if dev.type == "cuda":
            torch.cuda.synchronize()
    finally:
        lib.plan_destroy(plan)
    return y, gx, torch.view_as_complex(gw), gb.reshape(bias.shape) if bias is not None else gb, \
        torch.view_as_complex(xhat)


def rel_l2(a, b):
    a = np.asarray(a)
    b = np.asarray(b)
    return float(np.linalg.norm((a - b).ravel()) / max(np.linalg.norm(b.ravel()), 1e-300))


def _gemm(lib, a, b, out, P, Q, R, M, a_str, b_str, c_str, conj_a=False, conj_b=False, st=0):
    lib.modegemm(torch.view_as_real(a).data_ptr(), torch.view_as_real(b).data_ptr(),
                 torch.view_as_real(out).data_ptr(), st, P=P, Q=Q, R=R, n_modes=M,
                 a_sp=a_str[0], a_sr=a_str[1], a_sm=a_str[2], b_sr=b_str[0], b_sq=b_str[1], b_sm=b_str[2],
                 c_sp=c_str[0], c_sq=c_str[1], c_sm=c_str[2], conj_a=int(conj_a), conj_b=int(conj_b))


def staged_fwd_bwd(lib, x, w_block, bias, g, kept, out_spatial, freq_a=None, freq_s=None, real_col=0,
                   complex_data=False, fft_norm="forward", flags=0):
    """Forward + backward through the STAGE entry points with two plans: plan A (input grid, analysis
    map) for SC_FWD_SCALED / SC_INV_ADJ_R2C and plan B (output grid, synthesis map) for SC_INV_PADDED /
    SC_FWD_ADJ_C2R -- the call sequence neuraloperator_amd.spectral_conv issues when the output grid
    differs from the input's or the data is complex.  w_block: dense (Cin, Cout, *kept) complex.
    Returns (y, gx, gw_block, gbias)."""
    dev = x.device
    st = _stream(dev)
    b, cin = x.shape[:2]
    cout = w_block.shape[1]
    spatial = list(x.shape[2:])
    if complex_data:
        flags |= _lib.SC_PLAN_COMPLEX
    mk = int(np.prod(kept))
    pa = lib.plan_create(spatial, kept, fft_norm=fft_norm, flags=flags, freq=freq_a)
    pb = lib.plan_create(list(out_spatial), kept, fft_norm=fft_norm, flags=flags, freq=freq_s, real_col=real_col)
    try:
        cdt = torch.complex64
        xin = torch.view_as_real(x.contiguous()) if complex_data else x.contiguous()
        xhat = torch.empty(b, cin, *kept, dtype=cdt, device=dev)
        ws_a = torch.empty(max(lib.plan_workspace_bytes(pa, b * max(cin, cout)), 256), dtype=torch.uint8, device=dev)
        ws_b = torch.empty(max(lib.plan_workspace_bytes(pb, b * max(cin, cout)), 256), dtype=torch.uint8, device=dev)
        lib.transform_forward(pa, _lib.SC_FWD_SCALED, _ptr(xin), _ptr(torch.view_as_real(xhat)), b * cin, _ptr(ws_a), st)
        w = w_block.contiguous()
        yhat = torch.empty(b, cout, *kept, dtype=cdt, device=dev)
        _gemm(lib, xhat, w, yhat, b, cout, cin, mk, (cin * mk, mk, 1), (cout * mk, mk, 1), (cout * mk, mk, 1), st=st)
        if complex_data:
            y = torch.empty(b, cout, *out_spatial, dtype=cdt, device=dev)
            yout, bias_flat = torch.view_as_real(y), None
        else:
            y = torch.empty(b, cout, *out_spatial, dtype=torch.float32, device=dev)
            yout, bias_flat = y, (None if bias is None else bias.reshape(-1).contiguous())
        lib.transform_inverse(pb, _lib.SC_INV_PADDED, _ptr(torch.view_as_real(yhat)), _ptr(bias_flat), cout,
                              _ptr(yout), b * cout, _ptr(ws_b), st)
        if complex_data and bias is not None:
            y = y + bias
        # backward
        gin = torch.view_as_real(g.contiguous()) if complex_data else g.contiguous()
        ghat = torch.empty(b, cout, *kept, dtype=cdt, device=dev)
        lib.transform_forward(pb, _lib.SC_FWD_ADJ_C2R, _ptr(gin), _ptr(torch.view_as_real(ghat)), b * cout, _ptr(ws_b), st)
        if complex_data:
            gb = g.real.sum(dim=[0] + list(range(2, g.ndim))).reshape(bias.shape) if bias is not None else None
        else:
            gb = torch.empty(cout, dtype=torch.float32, device=dev)
            lib.bias_grad(pb, _ptr(torch.view_as_real(ghat)), b, cout, _ptr(gb), st)
            gb = gb.reshape(bias.shape) if bias is not None else gb
        gxhat = torch.empty(b, cin, *kept, dtype=cdt, device=dev)
        _gemm(lib, ghat, w, gxhat, b, cin, cout, mk, (cout * mk, mk, 1), (mk, cout * mk, 1), (cin * mk, mk, 1),
              conj_b=True, st=st)
        gw = torch.empty_like(w)
        _gemm(lib, xhat, ghat, gw, cin, cout, b, mk, (mk, cin * mk, 1), (cout * mk, mk, 1), (cout * mk, mk, 1),
              conj_a=True, st=st)
        if complex_data:
            gx = torch.empty(b, cin, *spatial, dtype=cdt, device=dev)
            gxo = torch.view_as_real(gx)
        else:
            gx = torch.empty(b, cin, *spatial, dtype=torch.float32, device=dev)
            gxo = gx
        lib.transform_inverse(pa, _lib.SC_INV_ADJ_R2C, _ptr(torch.view_as_real(gxhat)), 0, cin, _ptr(gxo), b * cin,
                              _ptr(ws_a), st)
        if dev.type == "cuda":
            torch.cuda.synchronize()
    finally:
        lib.plan_destroy(pa)
        lib.plan_destroy(pb)
    return y, gx, gw, gb
