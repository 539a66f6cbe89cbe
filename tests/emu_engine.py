"""TEST INFRASTRUCTURE ONLY: run the drop-in module on CPU tensors through the host-emulation build of the engine.

The product refuses CPU tensors and loads only the HIP library (no fallback).  For the CPU tier this context manager
points the package at tests/emu/libsc_engine_emu.so (the same kernel sources compiled for the host, thread per lane)
and neutralises the three CUDA runtime touch points of neuraloperator_amd.engine (device guard, current stream,
device context).  With it the REAL caller of the plug-in boundary -- the verbatim reference FNO -- can drive
``neuraloperator_amd.SpectralConv`` where /root/reference exists (here), which is nowhere a GPU is."""
import contextlib

import torch

from engine_runner import emu_lib


@contextlib.contextmanager
def engine_on_emulation():
    from neuraloperator_amd import _lib, blocks, engine

    saved = (_lib._LIB, engine._require_gpu, engine._stream, torch.cuda.device, torch.cuda.current_device,
             dict(engine._PLANS), blocks._require_gpu, blocks._stream, blocks._on_engine)
    engine._PLANS.clear()
    _lib._LIB = emu_lib()
    engine._require_gpu = lambda *a, **k: None
    engine._stream = lambda: 0
    blocks._require_gpu = lambda *a, **k: None
    blocks._stream = lambda: 0
    blocks._on_engine = lambda t: True
    torch.cuda.device = lambda *a, **k: contextlib.nullcontext()
    torch.cuda.current_device = lambda: 0
    try:
        yield _lib._LIB
    finally:
        engine._PLANS.clear()                       # emulation plans must never reach the product library
        _lib._LIB, engine._require_gpu, engine._stream, torch.cuda.device, torch.cuda.current_device = saved[:5]
        engine._PLANS.update(saved[5])
        blocks._require_gpu, blocks._stream, blocks._on_engine = saved[6:9]
