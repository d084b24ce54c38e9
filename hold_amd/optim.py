"""Optimiser step on ONE flat fp32 bucket.

The reference's optimiser is ``Adam(eps=1e-8)`` with the per-frame pose tables at ``0.1 * lr`` and every other
parameter at ``lr`` (code/src/hold/hold.py:79-101), after Lightning's ``gradient_clip_val=0.5`` (global L2 norm,
code/train.py:30).  ``FlatAdam`` re-homes every trainable parameter of a HOLDNet -- and its gradient -- as views into
two contiguous device buffers ordered [pose tables | everything else], so that

* the data-parallel exchange is one ``all_reduce`` of the gradient buffer itself (no gather / scatter copies),
* clip + Adam are three launches (``hold_sumsq`` + finalise, ``hold_adam_step``) with the clip factor read from
  device memory -- no host synchronisation anywhere in the step,
* ``zero_grad`` is one memset.

Semantics vs ``torch.optim.Adam``: identical arithmetic per element (bias corrections in double on the host); a
parameter whose ``.grad`` torch would leave ``None`` (no gradient at all this step) is treated as having a zero
gradient, i.e. its moments decay and the bias-correction step is the global one -- HOLD's parameters all receive
gradients from step 0, where the two coincide (verified against torch.optim.Adam in
tests/test_train_targets_gpu.py::test_flat_adam_matches_torch_adam_with_clipping).
"""
from __future__ import annotations

import torch
import torch.distributed as dist

from . import config
from ._lib import call, ptr
from .loss import reduce_workspace


def split_params(net):
    """(pose-table parameters, main parameters), trainable only, in module order (hold.py:83-97)."""
    low, low_ids = [], set()
    for node in net.nodes.values():
        for p in node.params.parameters():
            if p.requires_grad and p.numel():
                low.append(p)
                low_ids.add(id(p))
    # zero-sized parameters (the object's lin_pose.weight is [8, 0]) stay outside the bucket
    main = [p for p in net.parameters() if p.requires_grad and p.numel() and id(p) not in low_ids]
    return low, main


class FlatAdam:
    def __init__(self, net, lr=5e-4, pose_lr_scale=0.1, betas=(0.9, 0.999), eps=1e-8, clip_norm=0.5, group=None):
        self.net, self.lr, self.pose_lr_scale = net, float(lr), float(pose_lr_scale)
        self.betas, self.eps, self.clip_norm, self.group = betas, float(eps), float(clip_norm or 0.0), group
        low, main = split_params(net)
        self.params = low + main
        assert self.params, "no trainable parameters"
        dev = self.params[0].device
        # every parameter starts on a 256-byte boundary of the bucket (the kernels read weights with 16-byte vector
        # loads and LDS DMA); the padding elements stay zero in all four buffers, so they are inert under Adam
        self.offsets, off = [], 0
        for p in self.params:
            if p is (main[0] if main else None):
                self.n_low = off
            self.offsets.append(off)
            off += (p.numel() + self.ALIGN - 1) // self.ALIGN * self.ALIGN
        if not main:
            self.n_low = off
        self.n = off
        self.n_params = sum(p.numel() for p in self.params)
        self.flat = torch.zeros(self.n, device=dev)
        self.grad = torch.zeros(self.n, device=dev)
        self.m = torch.zeros(self.n, device=dev)
        self.v = torch.zeros(self.n, device=dev)
        self.sumsq = torch.zeros(1, device=dev)
        with torch.no_grad():
            for p, off in zip(self.params, self.offsets):
                k = p.numel()
                assert p.dtype == torch.float32 and p.device == dev
                self.flat[off:off + k].copy_(p.reshape(-1))
                p.data = self.flat[off:off + k].view(p.shape)
                p.grad = self.grad[off:off + k].view(p.shape)
                # backward kernels that produce a parameter's gradient may ADD it into p.grad themselves instead of
                # returning it to autograd (hold_net._WeightNormFn): one launch for a whole net, no AccumulateGrad per tensor
                p._hold_bucket = True
        self.step_count = 0

    ALIGN = 64  # floats

    def zero_grad(self):
        self.grad.zero_()
        for p, off in zip(self.params, self.offsets):  # autograd accumulates in place into these views; re-attach any that were replaced
            if p.grad is None or p.grad.data_ptr() != self.grad.data_ptr() + 4 * off:
                p.grad = self.grad[off:off + p.numel()].view(p.shape)

    def gather_stray_grads(self):
        """fold gradients that autograd attached as fresh tensors (instead of accumulating into the bucket view) back in."""
        for p, off in zip(self.params, self.offsets):
            k = p.numel()
            if p.grad is not None and p.grad.data_ptr() != self.grad.data_ptr() + 4 * off:
                self.grad[off:off + k].add_(p.grad.reshape(-1))
                p.grad = self.grad[off:off + k].view(p.shape)

    def allreduce(self, average=True):
        """the ONE collective of a data-parallel step: RCCL all-reduce of the gradient bucket over xGMI."""
        if dist.is_available() and dist.is_initialized():
            dist.all_reduce(self.grad, op=dist.ReduceOp.SUM, group=self.group)
            return 1.0 / dist.get_world_size(self.group) if average else 1.0
        return 1.0

    def rehome(self):
        """re-attach parameters whose storage left the bucket (``net.to()`` / ``.float()`` / ``p.data = ...`` after
        construction, e.g. GenericParams.init_parameters): their current values are copied in -- without this step()
        would keep updating the bucket while the model reads the detached storage."""
        moved = 0
        with torch.no_grad():
            for p, off in zip(self.params, self.offsets):
                if p.data_ptr() != self.flat.data_ptr() + 4 * off:
                    k = p.numel()
                    self.flat[off:off + k].copy_(p.data.reshape(-1).to(self.flat.device, torch.float32))
                    p.data = self.flat[off:off + k].view(p.shape)
                    moved += 1
        return moved

    def state_dict(self):
        """optimiser state for checkpoint / resume (the reference's Lightning checkpoints carry Adam's moments and step
        count): per-parameter views are not needed, the layout is (offsets, numel) over the flat bucket."""
        return {"step_count": self.step_count, "m": self.m.clone(), "v": self.v.clone(), "offsets": list(self.offsets),
                "numels": [p.numel() for p in self.params], "n": self.n, "n_low": self.n_low, "lr": self.lr,
                "pose_lr_scale": self.pose_lr_scale, "betas": tuple(self.betas), "eps": self.eps, "clip_norm": self.clip_norm}

    def load_state_dict(self, sd):
        if list(sd["offsets"]) != list(self.offsets) or list(sd["numels"]) != [p.numel() for p in self.params]:
            raise ValueError("FlatAdam.load_state_dict: the parameter layout of the checkpoint differs from this model's")
        self.step_count = int(sd["step_count"])
        self.m.copy_(sd["m"].to(self.m.device))
        self.v.copy_(sd["v"].to(self.v.device))
        for k in ("lr", "pose_lr_scale", "eps", "clip_norm"):
            if k in sd:
                setattr(self, k, float(sd[k]))
        if "betas" in sd:
            self.betas = tuple(sd["betas"])

    def step(self, grad_mul=1.0):
        """reduce (if distributed) -> clip by global norm on the reduced gradients -> Adam."""
        self.rehome()
        self.gather_stray_grads()
        grad_mul = grad_mul * self.allreduce(average=True)
        self.step_count += 1
        ws = reduce_workspace(self.grad.device)
        if self.clip_norm > 0:
            call("hold_sumsq", ptr(self.grad), self.n, ptr(self.sumsq), 0, ptr(ws))
        call("hold_adam_step", ptr(self.flat), ptr(self.grad), ptr(self.m), ptr(self.v), self.n, self.n_low,
             self.lr * self.pose_lr_scale, self.lr, self.betas[0], self.betas[1], self.eps, self.step_count,
             float(grad_mul), self.clip_norm, ptr(self.sumsq) if self.clip_norm > 0 else None)
        config.bump_weights_epoch()  # parameters changed behind torch's version counters: cached weight packs are stale

    def grad_norm(self):
        """global L2 norm of the (reduced) gradient bucket -- a host read, for logging / tests only."""
        ws = reduce_workspace(self.grad.device)
        out = torch.zeros(1, device=self.grad.device)
        call("hold_sumsq", ptr(self.grad), self.n, ptr(out), 0, ptr(ws))
        return float(out.sqrt())
