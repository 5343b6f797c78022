"""`FusedAdam`: torch.optim.Adam's update (defaults: betas (0.9, 0.999), eps 1e-8, no weight decay, no amsgrad) with the
reference's element-wise gradient clamp in front of it (scripts/policy.py:250-253), for all parameter tensors of a model
in ONE kernel launch (csrc/k_train.hip: drlgx_adam_step) instead of a dozen framework kernels per step.

Same call surface as the torch optimiser the reference constructs (`zero_grad()`, `step()`, `state_dict()`); parameters
must be fp32 HIP tensors (at most 8 of them - the GCN has six).
"""
import ctypes as C

import torch
import torch.distributed as dist

from . import _lib


class GradientBucket(object):
    """The gradients of a list of parameters as VIEWS into one flat tensor, and the data-parallel exchange on it
    (one all-reduce of the flattened gradient per optimiser step): `start()` issues ONE in-place SUM
    all-reduce of the flat tensor without blocking (on NCCL/RCCL it runs on the backend's own stream), `finish()` makes the
    current stream wait for it and returns the factor 1 / world that turns the sum into the mean - for the optimiser to fold
    into its update (`FusedAdam.step(grad_scale=...)`) or, with `apply=True`, multiplied in here.  No concatenation, no
    copies back, nothing at all without a process group.  Works on any device (the multi-process tests run it over gloo)."""

    def __init__(self, params):
        self.params = [p for p in params]
        p0 = self.params[0]
        if any(p.dtype != p0.dtype or p.device != p0.device for p in self.params):
            raise ValueError("GradientBucket: every parameter must share one dtype and one device")
        # every slice starts on a 256-byte boundary: the HIP backward kernels write the gradients with 16-byte accesses
        self._align = max(1, 256 // p0.element_size())
        self._offsets = []
        n = 0
        for p in self.params:
            self._offsets.append(n)
            n += -(-p.numel() // self._align) * self._align
        self.flat = torch.zeros(n, dtype=p0.dtype, device=p0.device)
        self._views = [self.flat[off:off + p.numel()].view_as(p) for p, off in zip(self.params, self._offsets)]
        self.attach()
        self._work = None

    def attach(self):
        """(Re-)bind every parameter's .grad to its slice of the flat tensor (after a zero_grad(set_to_none=True)).  The
        slices are built once; a parameter that still holds its own is left alone (two calls per update on the fused path)."""
        for p, view in zip(self.params, self._views):
            g = p.grad
            if g is view:
                continue
            if g is not None and g.data_ptr() != view.data_ptr():
                view.copy_(g)
            p.grad = view

    @staticmethod
    def world(group=None):
        return dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1

    def start(self, group=None):
        if self._work is not None:  # (a start() without finish(), e.g. after an exception in between: settle it first)
            self._work.wait()
            self._work = None
        if self.world(group) > 1:
            self._work = dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=group, async_op=True)

    def finish(self, group=None, apply=False):
        ws = self.world(group)
        if self._work is not None:
            self._work.wait()
            self._work = None
        scale = 1.0 / ws
        if apply and ws > 1:
            self.flat.mul_(scale)
            return 1.0
        return scale


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
        self.bucket = GradientBucket(self.params)  # the gradients are views of ONE flat tensor: all-reduced in place
        self.step_count = 0
        self.param_groups = [{"params": self.params, "lr": self.lr, "betas": self.betas, "eps": self.eps}]

    def grads(self):
        """The gradient tensors (views of the bucket's flat tensor), in parameter order - `gcn_backward_raw` writes into them."""
        self.bucket.attach()
        return [p.grad for p in self.params]

    def zero_grad(self, set_to_none=False):
        for p in self.params:
            if p.grad is not None and set_to_none:
                p.grad = None
        if not set_to_none:
            self.bucket.attach()
            self.bucket.flat.zero_()

    def step(self, grad_scale=1.0):
        """grad_scale: factor on the gradient in front of the clamp (GradientBucket.finish: 1 / world size)."""
        n = len(self.params)
        self.step_count += 1
        vp = C.c_void_p
        arr = lambda ts: (vp * n)(*[t.data_ptr() for t in ts])  # noqa: E731
        grads = self.grads()
        sizes = (C.c_int64 * n)(*[p.numel() for p in self.params])
        dev = self.params[0].device
        stream = _lib.stream_ptr(dev)
        _lib.check(_lib.lib().drlgx_adam_step_scaled(vp(stream), n, arr([p.data for p in self.params]), arr(grads), arr(self.exp_avg),
                                                     arr(self.exp_avg_sq), sizes, self.param_groups[0]["lr"], self.betas[0],
                                                     self.betas[1], self.eps, self.step_count, self.grad_clamp, float(grad_scale)))

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
