"""hipGraph replay of one forward + backward step.

The engine's passes are plain kernel launches on PyTorch's current stream, so a whole step -- ``module(x)`` and the
autograd backward of its result -- records into one hipGraph (``torch.cuda.CUDAGraph``) and replays with ONE host
call.  It pays where the eager step is bound by the host's issue rate (~0.18 ms per SpectralConv step on the
round-3 box: a 2-D 64 x 64 grid with B = 64, C = 64 needs 0.146 ms of GPU time -- profiles/r03s2_graph_step_time.txt);
large grids are GPU-bound either way.  Results are bit-identical to the eager step (tests/test_gpu_graph.py).

A graph replays fixed addresses: ``x`` and ``grad_out`` are STATIC tensors (refill them in place with ``copy_``),
``step.output``, ``x.grad`` and every parameter's ``.grad`` are rewritten by each replay -- do not set them to None
between replays (``zero_grad(set_to_none=True)`` would detach the optimizer from the memory the graph writes).
Reference lines: the step this replays is spectral_convolution.py:419-571 and its autograd backward.
"""
import torch

__all__ = ["GraphedStep", "capture_step"]


class GraphedStep:
    """One captured forward + backward of ``module`` at ``x`` with the output gradient ``grad_out``."""

    def __init__(self, module, x, grad_out, warmup=3, post=None):
        """post: an optional callable recorded behind the backward pass (the mode-parallel layer's
        ``reduce_replicated_grads``: with the engine's native RCCL path -- mpu/rccl_native.py -- the exchanges and the
        gradient all-reduce record into the graph like any launch)."""
        if not (x.is_cuda and grad_out.is_cuda):
            raise RuntimeError("GraphedStep: hipGraph capture needs device tensors (there is no CPU path)")
        self.module, self.x, self.grad_out = module, x, grad_out
        self.params = [p for p in module.parameters() if p.requires_grad]
        # the capture resets every .grad and each replay OVERWRITES them: gradients accumulated before construction
        # would be dropped silently -- refuse instead
        held = [n for n, p in module.named_parameters() if p.requires_grad and p.grad is not None]
        if x.requires_grad and x.grad is not None:
            held.append("x")
        if held:
            raise RuntimeError("GraphedStep: .grad is already set on " + ", ".join(held[:4]) +
                               (" ..." if len(held) > 4 else "") + "; a captured step overwrites gradients (no "
                               "accumulation across replays): apply or clear them before capturing")
        side = torch.cuda.Stream(device=x.device)
        side.wait_stream(torch.cuda.current_stream(x.device))
        with torch.cuda.stream(side):                    # plans, tables and workspaces are created outside the capture
            for _ in range(max(int(warmup), 1)):
                self._clear()
                module(x).backward(grad_out)
                if post is not None:
                    post()
        torch.cuda.current_stream(x.device).wait_stream(side)
        self._clear()
        mode = _quiesce_process_groups()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph, capture_error_mode=mode):
            self.output = module(x)
            self.output.backward(grad_out)
            if post is not None:
                post()
        self.output = self.output.detach()
        # the tensors every replay writes: optimizers must keep pointing at exactly these
        self.grads = [p.grad for p in self.params]
        self.x_grad = x.grad if x.requires_grad else None

    def _clear(self):
        if self.x.requires_grad:
            self.x.grad = None
        for p in self.params:
            p.grad = None

    def replay(self):
        """Run the step again on the current contents of ``x`` / ``grad_out`` / the parameters; returns the (static)
        output tensor.  Gradients are OVERWRITTEN, not accumulated."""
        for p, g in zip(self.params, self.grads):
            if p.grad is not g:
                raise RuntimeError("GraphedStep.replay: a parameter's .grad is no longer the captured tensor "
                                   "(zero_grad(set_to_none=True) or a reassignment detached it from the memory the "
                                   "graph writes); use zero_grad(set_to_none=False) or restore step.grads")
        if self.x_grad is not None and self.x.grad is not self.x_grad:
            raise RuntimeError("GraphedStep.replay: x.grad is no longer the captured tensor")
        self.graph.replay()
        return self.output

    __call__ = replay


def _quiesce_process_groups():
    """Before a capture in a process that has used torch.distributed on the device; returns the capture error mode.

    What raced in round 5: ProcessGroupNCCL's watchdog THREAD polls the events of the collectives it
    still tracks (hipEventQuery); under the default *global* capture mode an event query from ANY thread of the process
    while a stream is capturing is an error, the watchdog turns it into an abort -- one probe child in a few died with
    SIGABRT out of Watchdog::run().  Round 5 slept 0.8 s so that the watchdog had retired its work before the capture
    began: a heuristic.  Round 6: the capture runs in *thread_local* mode -- only the capturing thread's own calls are
    checked, the watchdog's queries are legal whenever they come -- after a real drain (device synchronize: every
    collective this process enqueued has finished, so nothing the watchdog still holds can be touched by the captured
    work).  SC_GRAPH_QUIESCE_MS (default 0) adds a pause on top for stacks where that is not enough."""
    import os
    import time
    try:
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized()):
            return "global"
    except Exception:
        return "global"
    torch.cuda.synchronize()
    ms = float(os.environ.get("SC_GRAPH_QUIESCE_MS", "0"))
    if ms > 0:
        time.sleep(ms / 1e3)
    return "thread_local"


def capture_step(module, x, grad_out, warmup=3, post=None):
    return GraphedStep(module, x, grad_out, warmup, post)
