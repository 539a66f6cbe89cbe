"""AdamW on the engine: the optimizer of the reference trainer
(/root/reference/neuralop/training/adamw.py:11-200) with the whole per-parameter update in ONE launch
(sc_adamw_step) instead of ~10 elementwise ATen kernels -- for the 69 MB complex spectral weight of the
metric layer that is 7 arrays across HBM once instead of ~30.

Same constructor arguments, state-dict layout (``step``, ``exp_avg``, ``exp_avg_sq`` -- complex for complex
parameters, exactly like the reference's ``torch.zeros_like(grad)`` state) and arithmetic order as the
reference.  Tensor-GaLore (``galore_params``, adamw.py:94-111, 139-153, 183-185): those parameters form their own
group, their gradient is projected onto a Tucker subspace (neuraloperator_amd.galore.TensorGaLoreProjector), the moments
live in the low-rank space and the normalised update is projected back before it is applied.
The fused launch takes contiguous fp32 / complex64 parameters on the GPU -- the spectral weights and everything
else an FNO holds by default; any other parameter of the model (CPU, bf16 / fp16 / fp64, channels_last views) is
updated with the same formulas as elementwise torch operations, so one optimizer serves a whole model and a step
never stops half way."""
import math
from typing import Callable, Iterable, Tuple

import torch
from torch.optim import Optimizer

from . import _lib


class AdamW(Optimizer):
    def __init__(self, params: Iterable, lr: float = 1e-3, betas: Tuple[float, float] = (0.9, 0.999),
                 eps: float = 1e-6, weight_decay: float = 0.0, correct_bias: bool = True,
                 galore_params=None, galore_rank=1.0, galore_update_proj_gap: int = 50, galore_scale: float = 1.0,
                 activation_checkpoint: bool = False, warm_restart: bool = True):
        if lr < 0.0:
            raise ValueError(f"Invalid learning rate: {lr} - should be >= 0.0")            # adamw.py:73-80
        if not 0.0 <= betas[0] < 1.0:
            raise ValueError(f"Invalid beta parameter: {betas[0]} - should be in [0.0, 1.0)")
        if not 0.0 <= betas[1] < 1.0:
            raise ValueError(f"Invalid beta parameter: {betas[1]} - should be in [0.0, 1.0)")
        if not 0.0 <= eps:
            raise ValueError(f"Invalid epsilon value: {eps} - should be >= 0.0")
        defaults = {"lr": lr, "betas": betas, "eps": eps, "weight_decay": weight_decay, "correct_bias": correct_bias}
        super().__init__(params, defaults)
        if galore_params is not None:                                                         # adamw.py:94-106
            self.add_param_group({"params": galore_params, "rank": galore_rank, "lr": lr, "betas": betas, "eps": eps,
                                  "weight_decay": weight_decay, "correct_bias": correct_bias, "galore": True})
        self.galore_rank = galore_rank
        self.activation_checkpoint = activation_checkpoint
        self.warm_restart = warm_restart
        self.galore_update_proj_gap = galore_update_proj_gap
        self.galore_scale = galore_scale

    @torch.no_grad()
    def step(self, closure: Callable = None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        lib = None                                  # resolved when the first fused-eligible parameter is met: a model
        for group in self.param_groups:             # without one (CPU-only, bf16, GaLore-only) never needs the library
            beta1, beta2 = group["betas"]
            for p in group["params"]:
                if p.grad is None:
                    continue
                grad = p.grad
                if grad.is_sparse:
                    raise RuntimeError("Adam does not support sparse gradients, please consider SparseAdam instead")
                cplx = torch.is_complex(p)
                fused = p.is_cuda and p.dtype in (torch.float32, torch.complex64) and p.is_contiguous()
                state = self.state[p]
                if "step" not in state:
                    state["step"] = 0
                if group.get("galore", False):
                    self._galore_update(p, grad, state, group)
                    continue
                if not fused:
                    self._elementwise_update(p, grad, state, group)
                    continue
                grad = grad.to(p.dtype).contiguous()
                if "exp_avg" not in state:
                    state["exp_avg"] = torch.zeros_like(grad)
                    state["exp_avg_sq"] = torch.zeros_like(grad)
                state["step"] += 1
                m, v = state["exp_avg"], state["exp_avg_sq"]
                view = torch.view_as_real if cplx else (lambda t: t)
                if lib is None:
                    lib = _lib.get_lib()
                with torch.cuda.device(p.device):
                    lib.adamw_step(view(p).data_ptr(), view(grad).data_ptr(), view(m).data_ptr(), view(v).data_ptr(),
                                   p.numel(), cplx, torch.cuda.current_stream().cuda_stream,
                                   lr=group["lr"], beta1=beta1, beta2=beta2, eps=group["eps"],
                                   weight_decay=group["weight_decay"], correct_bias=group["correct_bias"],
                                   step=state["step"])
        return loss

    def _galore_update(self, p, grad, state, group):
        """adamw.py:139-196 with ``group["galore"]``: project, Adam moments in the low-rank space, project back."""
        from .galore import TensorGaLoreProjector
        if "projector" not in state:
            state["projector"] = TensorGaLoreProjector(
                rank=self.galore_rank, update_proj_gap=self.galore_update_proj_gap, scale=self.galore_scale,
                activation_checkpoint=self.activation_checkpoint, warm_restart=self.warm_restart)
        proj = state["projector"]
        g = proj.project(grad, state["step"])
        if "exp_avg" not in state:
            state["exp_avg"] = torch.zeros_like(g)
            state["exp_avg_sq"] = torch.zeros_like(g)
        m, v = state["exp_avg"], state["exp_avg_sq"]
        beta1, beta2 = group["betas"]
        state["step"] += 1
        m.mul_(beta1).add_(g, alpha=1.0 - beta1)
        v.mul_(beta2).addcmul_(g, g.conj() if torch.is_complex(g) else g, value=1.0 - beta2)
        step_size = group["lr"]
        if group["correct_bias"]:
            step_size *= math.sqrt(1.0 - beta2 ** state["step"]) / (1.0 - beta1 ** state["step"])
        p.add_(proj.project_back(m / v.sqrt().add_(group["eps"])), alpha=-step_size)
        if group["weight_decay"] > 0.0:
            p.add_(p, alpha=-group["lr"] * group["weight_decay"])

    @staticmethod
    def _elementwise_update(p, grad, state, group):
        """The same update (adamw.py:155-200) as torch operations, for parameters the fused launch does not take."""
        if "exp_avg" not in state:
            state["exp_avg"] = torch.zeros_like(grad)
            state["exp_avg_sq"] = torch.zeros_like(grad)
        m, v = state["exp_avg"], state["exp_avg_sq"]
        beta1, beta2 = group["betas"]
        state["step"] += 1
        m.mul_(beta1).add_(grad, alpha=1.0 - beta1)
        v.mul_(beta2).addcmul_(grad, grad.conj() if torch.is_complex(grad) else grad, value=1.0 - beta2)
        step_size = group["lr"]
        if group["correct_bias"]:
            step_size *= math.sqrt(1.0 - beta2 ** state["step"]) / (1.0 - beta1 ** state["step"])
        p.add_(m / v.sqrt().add_(group["eps"]), alpha=-step_size)
        if group["weight_decay"] > 0.0:
            p.add_(p, alpha=-group["lr"] * group["weight_decay"])
