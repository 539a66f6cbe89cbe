"""GPU tier: a SpectralConv forward + backward step replayed as one hipGraph (neuraloperator_amd/graph.py) gives
bit-identical results to the eager step, also after the static inputs were refilled in place."""
import pytest
import torch

pytestmark = pytest.mark.gpu
needs_gpu = pytest.mark.skipif(not torch.cuda.is_available(), reason="GPU tier: no GPU visible")


def _flat(t):
    return torch.view_as_real(t) if t.is_complex() else t


@needs_gpu
@pytest.mark.parametrize("spatial,n_modes,kw", [
    ((64, 64), (32, 32), {}),
    ((256, 256), (64, 64), {}),
    ((16, 64, 64), (8, 16, 16), {}),
    ((64, 64), (16, 16), dict(factorization="tucker", rank=0.5, implementation="factorized")),
])
def test_graph_replay_matches_eager(spatial, n_modes, kw):
    from neuraloperator_amd import SpectralConv
    from neuraloperator_amd.graph import capture_step
    dev = torch.device("cuda:0")
    torch.manual_seed(5)
    conv = SpectralConv(6, 10, n_modes, **kw).to(dev)
    x = torch.randn(3, 6, *spatial, device=dev, requires_grad=True)
    g = torch.randn(3, 10, *spatial, device=dev)
    step = capture_step(conv, x, g)
    params = [p for p in conv.parameters() if p.requires_grad]
    for trial in range(2):
        if trial:                                        # new values through the same addresses
            with torch.no_grad():
                x.copy_(torch.randn_like(x))
                g.copy_(torch.randn_like(g))
        y = step.replay().clone()
        got = [x.grad.clone()] + [p.grad.clone() for p in params]
        xe = x.detach().clone().requires_grad_(True)
        saved = [p.grad for p in params]
        for p in params:
            p.grad = None
        ye = conv(xe)
        ye.backward(g)
        want = [xe.grad] + [p.grad for p in params]
        for p, s in zip(params, saved):                  # the graph keeps writing the tensors it captured
            p.grad = s
        assert torch.equal(y, ye.detach())
        for a, b in zip(got, want):
            assert torch.equal(_flat(a), _flat(b))


def test_graph_needs_a_device():
    from neuraloperator_amd.graph import capture_step
    lin = torch.nn.Linear(2, 2)
    with pytest.raises(RuntimeError, match="no CPU path"):
        capture_step(lin, torch.zeros(1, 2), torch.zeros(1, 2))


@needs_gpu
@pytest.mark.parametrize("kw", [dict(factorization="tucker", rank=0.5, implementation="factorized"),
                                dict(factorization="cp", rank=0.5, implementation="factorized")], ids=["tucker", "cp"])
def test_small_factor_gradients_are_reproducible(kw):
    """Factor gradients too small for the matrix-core kernel used float atomics (arrival order: last bits differed
    from run to run); they now go through workspace slots and a fixed-order reduction (sc_modegemm_msum_ws, path 0)."""
    from neuraloperator_amd import SpectralConv
    dev = torch.device("cuda:0")
    torch.manual_seed(5)
    conv = SpectralConv(6, 10, (16, 16), **kw).to(dev)
    x = torch.randn(3, 6, 64, 64, device=dev)
    g = torch.randn(3, 10, 64, 64, device=dev)
    params = [p for p in conv.parameters() if p.requires_grad]
    runs = []
    for _ in range(4):
        for p in params:
            p.grad = None
        xe = x.clone().requires_grad_(True)
        conv(xe).backward(g)
        runs.append([xe.grad.clone()] + [p.grad.clone() for p in params])
    for other in runs[1:]:
        for a, b in zip(runs[0], other):
            assert torch.equal(_flat(a), _flat(b))


@needs_gpu
def test_capture_refuses_pending_gradients_and_detached_grads():
    """a captured step overwrites .grad -- construction refuses gradients that are already accumulated, and
    a replay refuses to run once an optimizer's zero_grad(set_to_none=True) detached a parameter from the captured
    gradient tensor."""
    from neuraloperator_amd import SpectralConv
    from neuraloperator_amd.graph import capture_step
    dev = torch.device("cuda:0")
    torch.manual_seed(1)
    conv = SpectralConv(4, 4, (8, 8)).to(dev)
    x = torch.randn(2, 4, 32, 32, device=dev, requires_grad=True)
    g = torch.randn(2, 4, 32, 32, device=dev)
    conv(x).backward(g)                                  # gradients accumulated by the caller
    with pytest.raises(RuntimeError, match="already set"):
        capture_step(conv, x, g)
    conv.zero_grad(set_to_none=True)
    x.grad = None
    step = capture_step(conv, x, g)
    step.replay()
    assert all(p.grad is gr for p, gr in zip(step.params, step.grads))
    conv.zero_grad(set_to_none=True)
    with pytest.raises(RuntimeError, match="no longer the captured tensor"):
        step.replay()
    for p, gr in zip(step.params, step.grads):           # restoring the captured tensors makes it valid again
        p.grad = gr
    step.replay()


@needs_gpu
def test_raw_stream_accessor_fallback(monkeypatch):
    """engine._stream() uses a private torch accessor for the raw hipStream_t; without it the public (slower) route
    must hand the C-ABI the same handle."""
    from neuraloperator_amd import engine
    fast = engine._stream()
    monkeypatch.delattr(torch._C, "_cuda_getCurrentRawStream")
    assert engine._stream() == fast == torch.cuda.current_stream().cuda_stream
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        assert engine._stream() == side.cuda_stream


@needs_gpu
@pytest.mark.parametrize("batch,kw", [(1, dict(comm_chunks=1, chunk_dim="channels")), (1, dict(comm_chunks=3, chunk_dim="channels")),
                                      (4, dict())], ids=["one_piece", "channel_slabs", "batch_chunks"])
def test_mode_parallel_step_with_native_rccl_records_into_a_graph(batch, kw):
    """Round 4 (mpu/rccl_native.py): with the exchanges issued by ncclAllToAll / grouped ncclSend + ncclRecv straight on HIP
    streams, the mode-parallel layer's whole step -- transforms, exchanges, contractions, the bias all-reduce -- records into
    ONE hipGraph.  One-rank RCCL group (the only multi-process RCCL set-up a 1-GPU box allows): a replay equals the eager
    native step bit for bit, also after the static inputs were refilled, and the native path's results equal the
    torch.distributed path's bit for bit.  The case runs in its own process (tests/native_rccl_graph_case.py says why)."""
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    out = subprocess.run([sys.executable, os.path.join(here, "native_rccl_graph_case.py"), str(batch), repr(kw)],
                         capture_output=True, text=True, timeout=180)
    assert out.returncode == 0 and "CASE OK" in out.stdout, (out.stdout[-1500:], out.stderr[-3000:])
