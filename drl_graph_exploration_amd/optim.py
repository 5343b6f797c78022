"""`FusedAdam`: torch.optim.Adam's update (defaults: betas (0.9, 0.999), eps 1e-8, no weight decay, no amsgrad) with the
reference's element-wise gradient clamp in front of it (scripts/policy.py:250-253), for all parameter tensors of a model
in ONE kernel launch (csrc/k_train.hip: drlgx_adam_step) instead of a dozen framework kernels per step.

Same call surface as the torch optimiser the reference constructs (`zero_grad()`, `step()`, `state_dict()`); parameters
must be fp32 HIP tensors (at most 8 of them - the GCN has six).
"""
import ctypes as C

import torch

from . import _lib


class FusedAdam(object):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, grad_clamp=0.0):
        self.params = [p for p in params]
        if not self.params or len(self.params) > 8:
            raise ValueError("FusedAdam takes 1..8 parameter tensors")
        for p in self.params:
            if not p.is_cuda or p.dtype != torch.float32 or not p.is_contiguous():
                raise _lib.DrlgxError("FusedAdam needs contiguous fp32 HIP tensors (no CPU fallback)")
        self.lr, self.betas, self.eps, self.grad_clamp = float(lr), (float(betas[0]), float(betas[1])), float(eps), float(grad_clamp)
        self.exp_avg = [torch.zeros_like(p) for p in self.params]
        self.exp_avg_sq = [torch.zeros_like(p) for p in self.params]
        self.step_count = 0
        self.param_groups = [{"params": self.params, "lr": self.lr, "betas": self.betas, "eps": self.eps}]

    def grads(self):
        """The gradient tensors (allocated on first use), in parameter order - `gcn_backward_raw` writes into them."""
        for p in self.params:
            if p.grad is None:
                p.grad = torch.zeros_like(p)
        return [p.grad for p in self.params]

    def zero_grad(self, set_to_none=False):
        for p in self.params:
            if p.grad is not None:
                if set_to_none:
                    p.grad = None
                else:
                    p.grad.zero_()

    def step(self):
        n = len(self.params)
        self.step_count += 1
        vp = C.c_void_p
        arr = lambda ts: (vp * n)(*[t.data_ptr() for t in ts])  # noqa: E731
        grads = self.grads()
        sizes = (C.c_int64 * n)(*[p.numel() for p in self.params])
        dev = self.params[0].device
        stream = torch.cuda.current_stream(dev).cuda_stream
        _lib.check(_lib.lib().drlgx_adam_step(vp(stream), n, arr([p.data for p in self.params]), arr(grads), arr(self.exp_avg),
                                              arr(self.exp_avg_sq), sizes, self.param_groups[0]["lr"], self.betas[0], self.betas[1],
                                              self.eps, self.step_count, self.grad_clamp))

    def state_dict(self):
        return {"step": self.step_count, "exp_avg": [t.clone() for t in self.exp_avg], "exp_avg_sq": [t.clone() for t in self.exp_avg_sq],
                "lr": self.param_groups[0]["lr"], "betas": self.betas, "eps": self.eps, "grad_clamp": self.grad_clamp}

    def load_state_dict(self, st):
        self.step_count = int(st["step"])
        for a, b in zip(self.exp_avg, st["exp_avg"]):
            a.copy_(b)
        for a, b in zip(self.exp_avg_sq, st["exp_avg_sq"]):
            a.copy_(b)
        self.param_groups[0]["lr"] = float(st["lr"])
