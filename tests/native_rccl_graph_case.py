"""Helper of tests/test_gpu_graph.py::test_mode_parallel_step_with_native_rccl_records_into_a_graph: one case, run in
its OWN process (a fresh RCCL / HIP state: on this stack a mode-parallel step ordered against torch's default stream makes
a LATER capture segfault in hipStreamEndCapture, and a crash or hang here must not take the GPU tier with it).
Usage: python tests/native_rccl_graph_case.py BATCH 'KWARGS-as-python-dict'"""
import ast
import os
import socket
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from neuraloperator_amd.graph import capture_step  # noqa: E402
from neuraloperator_amd.mpu import ModeParallelSpectralConv, comm, rccl_native  # noqa: E402


def _flat(t):
    return torch.view_as_real(t) if t.is_complex() else t


def main(batch, kw):
    port = comm.free_port()                               # outside the ephemeral range (EADDRINUSE otherwise, now and then)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    dev = torch.device("cuda:0")
    comm.init(model_parallel_size=1, backend="nccl")
    torch.manual_seed(3)
    conv = ModeParallelSpectralConv(6, 5, (16, 12), **kw).to(dev)
    x = torch.randn(batch, 6, 32, 24, device=dev, requires_grad=True)
    g = torch.randn(batch, 5, 32, 24, device=dev)
    params = [p for p in conv.parameters() if p.requires_grad]

    side = torch.cuda.Stream()

    def eager(on_side):
        # eager steps of the layer run under a NON-default stream here: on this stack (ROCm 7.0 / RCCL 2.26.6) a step
        # whose exchanges were ordered against torch's default (legacy null) stream makes a LATER capture of the layer
        # segfault in hipStreamEndCapture, whichever path issued them; the torch-path comparison step
        # therefore comes last
        xe = x.detach().clone().requires_grad_(True)
        for p in params:
            p.grad = None
        torch.cuda.synchronize()
        if on_side:
            with torch.cuda.stream(side):
                ye = conv(xe)
                ye.backward(g)
                conv.reduce_replicated_grads()
        else:
            ye = conv(xe)
            ye.backward(g)
            conv.reduce_replicated_grads()
        torch.cuda.synchronize()
        return [ye.detach().clone(), xe.grad.clone()] + [p.grad.clone() for p in params]

    rccl_native.prefer_native()
    with torch.cuda.stream(side):
        assert rccl_native.get(conv._group()) is not None, rccl_native.LAST_REASON
    torch.cuda.synchronize()
    step = capture_step(conv, x, g, post=conv.reduce_replicated_grads)
    native_eager = None
    for trial in range(2):
        if trial:
            with torch.no_grad():
                x.copy_(torch.randn_like(x))
                g.copy_(torch.randn_like(g))
        y = step.replay().clone()
        rep = [y, x.grad.clone()] + [p.grad.clone() for p in params]
        saved = [p.grad for p in params]
        native_eager = eager(True)
        for p, sv in zip(params, saved):
            p.grad = sv
        for a, b in zip(rep, native_eager):
            assert torch.equal(_flat(a), _flat(b))
    # the torch.distributed path on the same inputs: same bits
    del step
    torch.cuda.synchronize()
    rccl_native.shutdown()
    rccl_native.prefer_native(False)
    assert rccl_native.get(conv._group()) is None, "eager steps stay on torch.distributed unless asked"
    want = eager(False)
    for a, b in zip(native_eager, want):
        assert torch.equal(_flat(a), _flat(b))
    comm.cleanup()
    print("CASE OK")


if __name__ == "__main__":
    main(int(sys.argv[1]), ast.literal_eval(sys.argv[2]))
