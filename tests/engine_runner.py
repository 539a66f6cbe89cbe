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
    """Build (if stale) and load the host-emulation library.  TEST ONLY."""
    srcs = [os.path.join(ROOT, "neuraloperator_amd", "csrc", f)
            for f in os.listdir(os.path.join(ROOT, "neuraloperator_amd", "csrc"))
            if f.endswith((".h", ".cpp"))]
    srcs += [os.path.join(EMU_DIR, "sc_emu_runtime.cpp"), os.path.join(ROOT, "include", "sc_engine.h")]
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
        y = torch.empty(b, cout, *spatial, dtype=torch.float32, device=dev)
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
