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
        """re-attach parameters whose storage left the bucket (``.float()`` / ``p.data = ...`` after construction, e.g.
        GenericParams.init_parameters): their current values are copied in and their ``.grad`` views are re-attached --
        without this step() would keep updating the bucket while the model reads the detached storage.  A parameter that
        was moved to ANOTHER DEVICE (``net.to(other)`` after the optimiser was built) is an error: silently pulling it
        back would leave the model split over two devices -- build the optimiser after placing the model."""
        moved = 0
        with torch.no_grad():
            for p, off in zip(self.params, self.offsets):
                if p.data_ptr() != self.flat.data_ptr() + 4 * off:
                    if p.device != self.flat.device:
                        raise RuntimeError(f"FlatAdam: a parameter now lives on {p.device}, the bucket on {self.flat.device}: "
                                           "construct FlatAdam after moving the model")
                    k = p.numel()
                    self.flat[off:off + k].copy_(p.data.reshape(-1).to(torch.float32))
                    p.data = self.flat[off:off + k].view(p.shape)
                    stray = p.grad if (p.grad is not None and p.grad.data_ptr() != self.grad.data_ptr() + 4 * off) else None
                    p.grad = self.grad[off:off + k].view(p.shape)
                    if stray is not None and stray.device == self.grad.device:
                        p.grad.add_(stray.reshape(p.shape))
                    p._hold_bucket = True
                    moved += 1
        return moved

    # ---- interchange with torch.optim.Adam (what the reference's Lightning checkpoints hold: per-parameter `step`,
    # `exp_avg`, `exp_avg_sq`, code/src/hold/hold.py:79-101).  The reference builds its parameter groups from python sets,
    # so a checkpoint's parameter INDICES carry no stable meaning; the exchange is therefore by parameter identity (a
    # torch optimiser built over this model's parameters) or by parameter name.
    def named_state(self):
        """{parameter name: {"step", "exp_avg", "exp_avg_sq"}} -- Adam's per-parameter state as torch.optim.Adam keeps it"""
        names = {id(p): n for n, p in self.net.named_parameters()}
        out = {}
        for p, off in zip(self.params, self.offsets):
            k = p.numel()
            out[names[id(p)]] = {"step": torch.tensor(float(self.step_count)),
                                 "exp_avg": self.m[off:off + k].view(p.shape).clone(),
                                 "exp_avg_sq": self.v[off:off + k].view(p.shape).clone()}
        return out

    def load_named_state(self, state):
        """inverse of named_state().  Validated BEFORE anything is written (a failed load leaves the optimiser untouched):
        every entry that is present must have the parameter's shape and all present entries the same step count; a bucket
        parameter WITHOUT an entry (torch.optim.Adam creates a parameter's state at its first gradient) gets zero moments
        under that common step -- what FlatAdam's own semantics give a parameter that never received a gradient."""
        names = {id(p): n for n, p in self.net.named_parameters()}
        steps, todo = set(), []
        for p, off in zip(self.params, self.offsets):
            st = state.get(names[id(p)])
            if st is None or not st:
                todo.append((off, p.numel(), None, None))
                continue
            for key in ("step", "exp_avg", "exp_avg_sq"):
                if key not in st:
                    raise KeyError(f"FlatAdam.load_named_state: {names[id(p)]} has no {key!r}")
            if tuple(st["exp_avg"].shape) != tuple(p.shape) or tuple(st["exp_avg_sq"].shape) != tuple(p.shape):
                raise ValueError(f"FlatAdam.load_named_state: state of {names[id(p)]} has shape {tuple(st['exp_avg'].shape)}, "
                                 f"the parameter {tuple(p.shape)}")
            steps.add(int(st["step"]))
            todo.append((off, p.numel(), st["exp_avg"], st["exp_avg_sq"]))
        if len(steps) > 1:
            raise ValueError(f"FlatAdam keeps ONE step count for all parameters; the given state has {sorted(steps)}")
        for off, k, m, v in todo:
            if m is None:
                self.m[off:off + k].zero_()
                self.v[off:off + k].zero_()
            else:
                self.m[off:off + k].copy_(m.reshape(-1).to(self.m.device))
                self.v[off:off + k].copy_(v.reshape(-1).to(self.v.device))
        self.step_count = steps.pop() if steps else 0

    def reference_groups(self):
        """the parameter groups of `HOLD.configure_optimizers` (code/src/hold/hold.py:79-101): ONE group per node with ALL of
        node.params.parameters() at 0.1 lr -- frozen and zero-sized parameters included, as the reference includes them --
        then one group with every remaining parameter of the model at lr.  Group COUNT and group SIZES are therefore those
        of a reference checkpoint's `optimizer_states[0]["param_groups"]`.  Order inside a node group: module order here; the
        reference builds it from a python `set`, so the index -> parameter assignment inside a node group of a reference
        checkpoint is whatever that process's hashes gave (see load_reference_state)."""
        groups, node_ids = [], set()
        for node in self.net.nodes.values():
            ps = list(node.params.parameters())
            node_ids.update(id(p) for p in ps)
            groups.append({"params": ps, "lr": self.lr * self.pose_lr_scale})
        main = [p for p in self.net.parameters() if id(p) not in node_ids]
        if main:
            groups.append({"params": main, "lr": self.lr})
        return groups

    def torch_optimizer(self):
        """a torch.optim.Adam with the reference's group structure (reference_groups) carrying this optimiser's state: what
        `HOLD.configure_optimizers` would hold at this point.  Parameters outside the bucket (frozen, zero-sized) are in
        their groups without state, as in the reference before their first gradient."""
        opt = torch.optim.Adam(self.reference_groups(), lr=self.lr, betas=tuple(self.betas), eps=self.eps)
        self.export_to(opt)
        return opt

    def export_to(self, opt):
        """write Adam's moments and step count into a torch optimiser built over the same parameter objects"""
        for p, off in zip(self.params, self.offsets):
            k = p.numel()
            opt.state[p] = {"step": torch.tensor(float(self.step_count)),
                            "exp_avg": self.m[off:off + k].view(p.shape).clone(),
                            "exp_avg_sq": self.v[off:off + k].view(p.shape).clone()}

    def import_from(self, opt):
        """read them back from a torch optimiser over this model's parameter objects (parameters torch holds no state for
        yet get zero moments)"""
        names = {id(p): n for n, p in self.net.named_parameters()}
        self.load_named_state({names[id(p)]: opt.state[p] for p in self.params if p in opt.state and opt.state[p]})

    def load_reference_state(self, opt_state_dict, ambiguous="raise", index_to_name=None):
        """Adam state of a REFERENCE checkpoint (`ckpt["optimizer_states"][0]`: {"state": {index: {...}}, "param_groups":
        [{"params": [indices], ...}, ...]}) -> this optimiser.  The group structure must be the reference's (one group per
        node, then the main group: reference_groups()); the main group's order is `model.parameters()` order, the same
        here, so its entries map by position.  Inside a NODE group the reference's order comes from a python set: entries
        are matched to that node's parameters by SHAPE, and where two parameters of a node share a shape (MANO's
        global_orient / transl tables, both [n_frames, 3] and both trained: EVERY real reference checkpoint) and carry
        different state the assignment is ambiguous.  What then happens is the caller's choice:
          ambiguous="raise" (default)  ValueError -- nothing is loaded;
          ambiguous="zero"             those tables start from zero moments (Adam's bias correction keeps running from the
                                       checkpoint's step count: their first updates are damped, not wrong), everything
                                       else -- all network weights, the unambiguous tables -- is loaded; a warning names them;
          index_to_name={index: name}  the caller knows the order (e.g. recorded `id()` order of the saving process, or
                                       recovered by matching exp_avg against the tables' sparsity pattern): explicit
                                       assignment of checkpoint indices to parameter names, applied before shape matching.
        Model WEIGHTS of a reference checkpoint are not affected by any of this (state_dict names are the reference's)."""
        groups = self.reference_groups()
        pg = opt_state_dict["param_groups"]
        if len(pg) != len(groups) or any(len(a["params"]) != len(b["params"]) for a, b in zip(pg, groups)):
            raise ValueError("FlatAdam.load_reference_state: group structure differs: checkpoint "
                             f"{[len(a['params']) for a in pg]}, model {[len(b['params']) for b in groups]}")
        names = {id(p): n for n, p in self.net.named_parameters()}
        state, named = opt_state_dict["state"], {}
        n_nodes = len(self.net.nodes)
        for gi, (a, b) in enumerate(zip(pg, groups)):
            if gi >= n_nodes:  # main group: positional
                for idx, p in zip(a["params"], b["params"]):
                    if idx in state and state[idx]:
                        named[names[id(p)]] = state[idx]
                continue
            taken = set()
            for idx in a["params"]:
                if index_to_name and idx in index_to_name and idx in state and state[idx]:
                    named[index_to_name[idx]] = state[idx]
                    taken.add(idx)
            entries = [state[idx] for idx in a["params"] if idx in state and state[idx] and idx not in taken]
            by_shape = {}
            for p in b["params"]:
                if names[id(p)] not in named:
                    by_shape.setdefault(tuple(p.shape), []).append(p)
            for shape, ps in by_shape.items():
                cand = [e for e in entries if tuple(e["exp_avg"].shape) == shape]
                if not cand:
                    continue
                if len(ps) == 1 and len(cand) == 1:
                    named[names[id(ps[0])]] = cand[0]
                elif all(torch.equal(c["exp_avg"], cand[0]["exp_avg"]) and torch.equal(c["exp_avg_sq"], cand[0]["exp_avg_sq"])
                         for c in cand) and len(cand) == len(ps):
                    for p_, c in zip(ps, cand):
                        named[names[id(p_)]] = c
                elif ambiguous == "zero":
                    import warnings
                    warnings.warn("FlatAdam.load_reference_state: Adam moments of " + ", ".join(names[id(p_)] for p_ in ps) +
                                  " start from zero (same shape, different state: the checkpoint's order inside a node group is "
                                  "a python set's)")
                else:
                    raise ValueError(f"FlatAdam.load_reference_state: {len(ps)} parameters of node group {gi} share the shape "
                                     f"{shape}; the reference orders a node group by a python set, the assignment is ambiguous "
                                     "(ambiguous='zero' loads everything else; index_to_name= resolves it)")
        self.load_named_state(named)

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
