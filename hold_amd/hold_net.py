"""HOLDNet on the HIP kernels: the host-side mirror of the reference's scene graph
(code/src/hold/hold_net.py:23-134, code/src/model/renderables/{node,mano_node,object_node,background}.py).

Module / parameter names equal the reference's, so its checkpoints load with ``load_state_dict``:
``nodes.<id>.implicit_network.lin<l>.{weight_g,weight_v,bias}``, ``nodes.<id>.rendering_network.{lin_pose,lin<l>}``,
``nodes.<id>.density.beta``, ``nodes.<id>.params.<name>.weight``, ``background.*``.
``forward(input)`` takes the same input dict and returns the same output keys.  All per-ray / per-sample
arithmetic runs in libholdhip.so; torch is used for parameters, per-frame (B x 16 x 4 x 4) algebra,
the autograd graph between kernels, and output bookkeeping.
"""
from __future__ import annotations

import math

import ctypes as C

import numpy as np

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import config
from . import field as _field
from . import gemm as G
from . import kernels as K
from .field import FEAT, FieldSpec, NodeField, Pool, pack_weights, pad4, zeros_like_many
from .fitting import seal_mano_mesh
from .geometry import MeshIndex, PointInSpace, compute_mano_cano_sdf, sample_on_barycentric_mesh, subdivide_loop
from .mano import MANOServer, ObjectServer
from .sampler import ErrorBoundSampler, UniformSampler
from .xdict import output_class

CLASS_ID = {"object": 1, "right": 2, "left": 3}


# ------------------------------------------------------------------------------------------ small modules
class Embedder(nn.Module):
    """plain Fourier embedding bookkeeping (embedders.py:7-50); the arithmetic is in hold_embed_fwd."""

    def __init__(self, input_dims, num_freq):
        super().__init__()
        self.input_dims, self.num_freq = input_dims, num_freq
        self.out_dim = input_dims + 2 * num_freq * input_dims

    def step(self):
        pass

    def eval(self):  # reference semantics: Embedder.eval() is a no-op (embedders.py:45-46)
        pass

    def weights(self, device):
        return None


class BarfEmbedder(Embedder):
    """coarse-to-fine mask of embedders.py:53-125 (alpha table, step counter buffers)."""

    def __init__(self, input_dims, num_freq, start, end, no_barf=False):
        super().__init__(input_dims, num_freq)
        self.no_barf, self.start, self.end = no_barf, start, end
        self.alphas = torch.cat((torch.zeros(start), torch.linspace(0, num_freq, end - start)), 0)
        self.register_buffer("alpha_iter", torch.tensor(0))
        self.register_buffer("alpha_max_iter", torch.tensor(len(self.alphas)))
        self._iter_host = 0  # host mirror of alpha_iter: step() must not read a device buffer back (a sync per node and step)
        self._register_load_state_dict_pre_hook(self._sync_host_counter)
        self.populate(self.alphas[int(self.alpha_iter)])

    def _sync_host_counter(self, state_dict, prefix, *args):
        v = state_dict.get(prefix + "alpha_iter")
        if v is not None:
            self._iter_host = int(v)
            self.populate(self.alphas[min(self._iter_host, len(self.alphas) - 1)])

    def populate(self, alpha):
        k = torch.arange(self.num_freq, dtype=torch.float32)
        ak = alpha - k
        w = torch.clamp(ak, 0, 1)
        ci = torch.logical_and(0 <= ak, ak < 1)
        cv = (1 - torch.cos(ak * math.pi)) / 2
        w[ci] = cv[ci]
        w = w[:, None].repeat(1, self.input_dims * 2).view(-1)
        self.barf_weights = torch.cat((torch.ones(self.input_dims), w), 0)

    def step(self):
        self._iter_host = min(self._iter_host + 1, len(self.alphas) - 1)
        self.alpha_iter.fill_(self._iter_host)  # the checkpointed buffer follows, without a device-to-host read
        self.populate(self.alphas[self._iter_host])

    def eval(self):  # embedders.py:124-125
        self.no_barf = True

    def weights(self, device):
        if self.no_barf:
            return None
        key = (str(device), self._iter_host)
        if getattr(self, "_w_dev", (None, None))[0] != key:  # one upload per schedule step, not one per query
            self._w_dev = (key, _lib_h2d(self.barf_weights.to(torch.float32).contiguous(), device))
        return self._w_dev[1]


def _lin(inn, out, weight_norm):
    l = nn.Linear(inn, out)
    return nn.utils.weight_norm(l) if weight_norm else l


def _pack_key(modules, training):
    """identity of the parameter values a weight pack was built from (see hold_amd/config.py: weight-pack invalidation)"""
    return (config.precision(), bool(training), config.weights_epoch(),
            tuple((p._version, p.data_ptr()) for m in modules for p in m.parameters()))


def _lib_h2d(t, device):
    from ._lib import h2d
    return h2d(t, device)


def _eff(lin):
    """effective weight of a (possibly weight-normed) Linear, differentiable w.r.t. weight_g / weight_v."""
    if hasattr(lin, "weight_g"):
        v, g = lin.weight_v, lin.weight_g
        return v * (g / v.norm(dim=1, keepdim=True))
    return lin.weight


class _WeightNormFn(torch.autograd.Function):
    """w_l = v_l * (g_l / ||v_l||_row) for ALL weight-normed layers of a net: one launch per direction
    (hold_weight_norm_fwd / _bwd, csrc/wnorm.hip) where autograd's per-layer graph costs ~10 tiny kernels per layer and
    direction.  Backward: t = rowsum(D * v), n = ||v||: dg = t / n, dv = D g / n - v t g / n^3.

    Parameters that live in FlatAdam's gradient bucket (``p._hold_bucket``, set by FlatAdam) get their gradients ADDED into
    ``p.grad`` by the kernel itself and ``None`` is returned to autograd for them -- no per-parameter AccumulateGrad
    launch; ``torch.autograd.grad`` w.r.t. such a parameter therefore sees None (use ``.backward()`` / ``p.grad``)."""

    @staticmethod
    def forward(ctx, L, *vg):
        from . import _lib
        vs, gs = vg[:L], vg[L:]
        dev = vs[0].device
        assert L <= 16 and all(v.is_cuda and v.is_contiguous() and v.dtype == torch.float32 for v in vs), \
            "weight normalisation runs on the HIP kernels only (hold_amd has no CPU path)"
        sizes = [(v.numel() + 63) // 64 * 64 for v in vs]
        flat = torch.empty(sum(sizes), device=dev)  # every effective matrix 256-byte aligned in ONE allocation
        outs, off = [], 0
        d = _lib.WnDesc()
        d.n_layers, d.accumulate = L, 0
        for i, (v, g) in enumerate(zip(vs, gs)):
            w = flat[off:off + v.numel()].view(v.shape)
            off += sizes[i]
            outs.append(w)
            l = d.layers[i]
            l.v, l.g, l.w = v.data_ptr(), g.data_ptr(), w.data_ptr()
            l.rows, l.cols, l.ldv, l.ldw = v.shape[0], v.shape[1], v.shape[1], v.shape[1]
        _lib.call("hold_weight_norm_fwd", C.byref(d))
        ctx.save_for_backward(*vs, *gs)
        ctx.L = L
        ctx.set_materialize_grads(False)
        return tuple(outs)

    @staticmethod
    def backward(ctx, *dW):
        from . import _lib
        L = ctx.L
        sv = ctx.saved_tensors
        vs, gs = sv[:L], sv[L:]
        # inputs: (L, v_0..v_{L-1}, g_0..g_{L-1}) -- a layer whose v AND g need no gradient (frozen after FlatAdam was built,
        # or excluded by backward(inputs=...)) is skipped altogether: the in-place path must not write gradients autograd
        # was told not to produce (round-3 advisor)
        need = [ctx.needs_input_grad[1 + i] or ctx.needs_input_grad[1 + L + i] for i in range(L)]
        direct = [dW[i] is not None and need[i] and getattr(vs[i], "_hold_bucket", False) and vs[i].grad is not None
                  and getattr(gs[i], "_hold_bucket", False) and gs[i].grad is not None
                  and ctx.needs_input_grad[1 + i] and ctx.needs_input_grad[1 + L + i] for i in range(L)]
        dv, dg = [None] * L, [None] * L
        keep = []  # contiguous copies of incoming gradients stay alive until their launch has been issued
        for mode in (True, False):  # one launch for the layers accumulated in place, one for those returned to autograd
            idx = [i for i in range(L) if dW[i] is not None and need[i] and direct[i] == mode]
            if not idx:
                continue
            d = _lib.WnDesc()
            d.n_layers, d.accumulate = len(idx), int(mode)
            if not mode:  # results carved out of one allocation, 256-byte aligned pieces
                sizes = [(vs[i].numel() + 63) // 64 * 64 for i in idx]
                flat = torch.empty(sum(sizes) + sum((vs[i].shape[0] + 63) // 64 * 64 for i in idx), device=vs[0].device)
                off = 0
            for j, i in enumerate(idx):
                v, g, D = vs[i], gs[i], dW[i].contiguous()
                keep.append(D)
                l = d.layers[j]
                if mode:
                    tv, tg = v.grad, g.grad
                else:
                    tv = flat[off:off + v.numel()].view(v.shape)
                    off += sizes[j]
                    tg = flat[off:off + v.shape[0]].view(g.shape)
                    off += (v.shape[0] + 63) // 64 * 64
                    dv[i], dg[i] = tv, tg
                assert tv.is_contiguous() and tg.is_contiguous()
                l.v, l.g, l.dw, l.dv, l.dg = v.data_ptr(), g.data_ptr(), D.data_ptr(), tv.data_ptr(), tg.data_ptr()
                l.rows, l.cols, l.ldv, l.ldw = v.shape[0], v.shape[1], v.shape[1], v.shape[1]
            _lib.call("hold_weight_norm_bwd", C.byref(d))
        return (None, *dv, *dg)


def _effective_all(lins):
    """effective weights of a list of Linears: one _WeightNormFn call for the weight-normed ones"""
    wn = [i for i, l in enumerate(lins) if hasattr(l, "weight_g")]
    out = [None if hasattr(l, "weight_g") else l.weight for l in lins]
    if wn:
        ws = _WeightNormFn.apply(len(wn), *[lins[i].weight_v for i in wn], *[lins[i].weight_g for i in wn])
        for i, w in zip(wn, ws):
            out[i] = w
    return out


class ImplicitNet(nn.Module):
    """ImplicitNet of code/src/networks/shape_net.py:8-144: the reference's parameter names / shapes and its
    ``init: geometry`` scheme (:50-70, drawn in the same order from torch's global generator), with
    ``forward(input, cond)`` / ``gradient(x, cond)`` running on the HIP kernels (first- and second-order backward
    to the weights through ``NodeField``).  ``cond`` is accepted for signature parity: the 45 MANO pose columns are
    multiplied by zero in the reference (:104-106) and the object net has none, so it never changes the result;
    the background net (cond = frame latent, d_in 4) is evaluated through ``Background`` only."""

    def __init__(self, d_in, multires, cond_dim, weight_norm, embedding="fourier", barf_s=1000, barf_e=10000,
                 no_barf=False, init="none", bias=0.6, skip_in=(4,)):
        super().__init__()
        if embedding == "barf":
            self.embedder_obj = BarfEmbedder(d_in, multires, barf_s, barf_e, no_barf)
        else:
            self.embedder_obj = Embedder(d_in, multires)
        e = self.embedder_obj.out_dim
        self.d_in, self.multires, self.cond_dim, self.E = d_in, multires, cond_dim, e
        dims = [e] + [256] * 8 + [1 + FEAT]
        self.num_layers = len(dims)
        self.skip_in = tuple(skip_in)
        for l in range(self.num_layers - 1):
            out = dims[l + 1] - dims[0] if (l + 1) in self.skip_in else dims[l + 1]
            lin = nn.Linear(dims[l] + (cond_dim if l == 0 else 0), out)
            if init == "geometry":
                if l == self.num_layers - 2:
                    nn.init.normal_(lin.weight, mean=np.sqrt(np.pi) / np.sqrt(dims[l]), std=0.0001)
                    nn.init.constant_(lin.bias, -bias)
                elif multires > 0 and l == 0:
                    nn.init.constant_(lin.bias, 0.0)
                    nn.init.constant_(lin.weight[:, 3:], 0.0)
                    nn.init.normal_(lin.weight[:, :3], 0.0, np.sqrt(2) / np.sqrt(out))
                elif multires > 0 and l in self.skip_in:
                    nn.init.constant_(lin.bias, 0.0)
                    nn.init.normal_(lin.weight, 0.0, np.sqrt(2) / np.sqrt(out))
                    nn.init.constant_(lin.weight[:, -(dims[0] - 3):], 0.0)
                else:
                    nn.init.constant_(lin.bias, 0.0)
                    nn.init.normal_(lin.weight, 0.0, np.sqrt(2) / np.sqrt(out))
            elif init == "zero" and l == self.num_layers - 2:
                nn.init.constant_(lin.bias, 0.0)
                nn.init.uniform_(lin.weight, -1e-5, 1e-5)
            setattr(self, f"lin{l}", nn.utils.weight_norm(lin) if weight_norm else lin)
        self._fields = {}
        self._pack_cache = (None, None)

    def effective(self):
        lins = [getattr(self, f"lin{l}") for l in range(9)]
        return _effective_all(lins), [l.bias for l in lins]

    def _pack(self, spec, iw, ib):
        """trunk-only weight pack for the kernel-backed call surface, rebuilt when a parameter changes"""
        key = _pack_key((self,), True)
        if self._pack_cache[0] != key:
            with torch.no_grad():
                self._pack_cache = (key, pack_weights(spec, iw, ib, None, None, need_bwd=True))
        return self._pack_cache[1]

    # ---- kernel-backed call surface (shape_net.py:84-144) ----
    def _field(self, device, tag):
        assert self.d_in == 3, "the background ImplicitNet (d_in 4, frame cond) is evaluated by Background.render"
        key = (str(device), tag)
        if key not in self._fields:
            self._fields[key] = NodeField(FieldSpec("hand" if self.cond_dim else "object"), device)
        return self._fields[key]

    def forward(self, input, cond=None, current_epoch=None):
        """input [B,P,3] | [P,3] -> [B,P,257] (sdf | 256 features), differentiable w.r.t. the weights and input."""
        if input.dim() == 2:
            input = input.unsqueeze(0)
        B, P, _ = input.shape
        if B * P == 0:
            return input
        iw, ib = self.effective()
        out = _ImplicitFn.apply(self, input.reshape(B * P, 3), self.embedder_obj.weights(input.device), *iw, *ib)
        return out.view(B, P, 1 + FEAT)

    @torch.no_grad()
    def sdf(self, x):
        """no-grad canonical SDF at x [P,3] -> [P] through the fused LDS-resident trunk (hold_fused_sdf): the query of
        the canonical meshing grid (hold_utils.query_oc under generate_mesh, code/src/utils/meshing.py:35-41)."""
        P = x.shape[0]
        fld = self._field(x.device, "oc")
        iw, ib = self.effective()
        pk = self._pack(fld.spec, iw, ib)
        xc = torch.zeros(P, 4, device=x.device)
        xc[:, :3] = x
        out = torch.empty(P, 1, device=x.device)
        wpack, bias8 = pk["fused"]
        K.fused_sdf(xc, P, wpack, bias8, pk["w8_sdf"], float(pk["b8_sdf"]), self.embedder_obj.weights(x.device), out)
        return out.view(P)

    def gradient(self, x, cond=None):
        """d sdf / d x with a differentiable (second-order) graph to the weights (shape_net.py:132-144): [P,1,3]."""
        iw, ib = self.effective()
        xc = torch.zeros(x.shape[0], 4, device=x.device)
        xc[:, :3] = x.detach()
        g = _EikonalFn.apply(self, xc, self.embedder_obj.weights(x.device), *iw, *ib)
        return g.unsqueeze(1)


class RenderingNet(nn.Module):
    """texture_net.py:8-44 parameter container."""

    def __init__(self, mode, d_in0, hidden, weight_norm, pose_dim=0):
        super().__init__()
        self.mode = mode
        if mode == "pose":
            self.lin_pose = nn.Linear(pose_dim, 8)
        dims = [d_in0] + list(hidden) + [3]
        self.num_layers = len(dims)
        for l in range(len(dims) - 1):
            setattr(self, f"lin{l}", _lin(dims[l], dims[l + 1], weight_norm))

    def effective(self):
        lins = [getattr(self, f"lin{l}") for l in range(self.num_layers - 1)]
        return _effective_all(lins), [l.bias for l in lins]


class LaplaceDensity(nn.Module):
    def __init__(self, beta=0.1, beta_min=1e-4):
        super().__init__()
        self.beta = nn.Parameter(torch.tensor(beta))
        self.beta_min = beta_min

    def get_beta(self):
        return self.beta.abs() + self.beta_min

    def beta_host(self):
        """the value of get_beta() as a Python float (kernel scalar arguments, the sampler's convergence test): ONE device
        read per parameter update instead of one per use -- every read drains the stream"""
        key = (config.weights_epoch(), self.beta._version, self.beta.data_ptr())
        if getattr(self, "_host", (None, None))[0] != key:
            self._host = (key, float(self.beta.detach().abs() + self.beta_min))
        return self._host[1]


class GenericParams(nn.Module):
    """per-frame pose tables (code/src/model/generic/params.py:6-62, mano/params.py:5-46, obj/params.py:4-30)."""

    def __init__(self, num_frames, params_dim, node_id):
        super().__init__()
        self.num_frames, self.params_dim, self.node_id = num_frames, params_dim, node_id
        self.param_names = list(params_dim.keys())
        for name, dim in params_dim.items():
            emb = nn.Embedding(1 if name == "betas" else num_frames, dim)
            emb.weight.data.fill_(0)
            emb.weight.requires_grad = False
            setattr(self, name, emb)

    def init_parameters(self, param_name, data, requires_grad=False):
        w = getattr(self, param_name).weight
        w.data = data[..., :self.params_dim[param_name]].to(device=w.device, dtype=w.dtype)
        w.requires_grad = requires_grad

    def set_requires_grad(self, param_name, requires_grad=True):
        getattr(self, param_name).weight.requires_grad = requires_grad

    def load_entity(self, data):
        """fill the tables from one ``entities[node_id]`` record of data.npy (docs/data_doc.md:70-87)."""
        t = lambda a: torch.tensor(np.asarray(a), dtype=torch.float32)
        if "pose" in self.param_names:  # MANOParams.load_params (mano/params.py:15-46)
            poses = np.asarray(data["hand_poses"])
            vals = {"betas": t(np.asarray(data["mean_shape"])[None]), "global_orient": t(poses[:, :3]),
                    "pose": t(poses[:, 3:]), "transl": t(data["hand_trans"])}
        else:  # ObjectParams.load_params (obj/params.py:9-30)
            op = np.asarray(data["object_poses"])
            vals = {"global_orient": t(op[:, :3]), "transl": t(op[:, 3:])}
        for name, v in vals.items():
            self.init_parameters(name, v, requires_grad=False)

    def load_params(self, case):
        import os
        ents = np.load(os.path.join("./data", case, "build/data.npy"), allow_pickle=True).item()["entities"]
        self.load_entity(ents[self.node_id])

    def forward(self, frame_ids):
        out = output_class()()
        for name in self.param_names:
            ids = torch.zeros_like(frame_ids) if name == "betas" else frame_ids
            out[f"{self.node_id}.{name}"] = getattr(self, name)(ids)
        if "pose" in self.param_names:
            out[f"{self.node_id}.full_pose"] = torch.cat(
                (out[f"{self.node_id}.global_orient"], out[f"{self.node_id}.pose"]), dim=1)
        return out

    def defrost(self, keys=None):
        for n in (keys or self.param_names):
            self.set_requires_grad(n, True)

    def freeze(self, keys=None):
        for n in (keys or self.param_names):
            self.set_requires_grad(n, False)


# ------------------------------------------------------------------------------------------ autograd glue
class _FieldFn(torch.autograd.Function):
    """sdf / colour / canonical normal of one node at given deformed-space points (fwd + hand-derived bwd)."""

    @staticmethod
    def forward(ctx, node, x, ppf, dfm_const, barf_w, training, tfs, pose_embed, time_code, *weights):
        pk = node._pk  # packed by Node.render from these same weights
        P = x.shape[0]
        nb = node.spec.n_bones
        dfm = dict(dfm_const)
        dfm["tfs"] = tfs.detach().reshape(-1, nb, 16).contiguous()
        out = node.field.forward(pk, x, P, ppf, dfm, barf_w, pose_embed.detach().contiguous(),
                                 None if time_code is None else time_code.detach().contiguous(), training=training,
                                 beta=node.density.beta_host())
        ctx.node, ctx.gen = node, node.field.gen
        ctx.B = tfs.shape[0]
        ctx.tfs_shape = tfs.shape
        ctx.has_time = time_code is not None
        sdf = out["sdf"].view(P)
        rgb = out["rgb"][:, :3]
        normal = out["normal"]
        xc = out["xc"][:, :3]
        ctx.mark_non_differentiable(xc)
        return sdf, rgb, normal, xc

    @staticmethod
    def backward(ctx, d_sdf, d_rgb, d_normal, d_xc_unused):
        node = ctx.node
        _check_gen(node.field, ctx.gen, f"HOLDNet node '{node.node_id}'")
        P = node.field.saved["P_full"]  # (the field itself may hold the compacted live samples only)
        dev = node.field.device
        d_sdf = torch.zeros(P, device=dev) if d_sdf is None else d_sdf.contiguous()
        d_rgb = torch.zeros(P, 3, device=dev) if d_rgb is None else d_rgb.contiguous()
        d_normal = None if d_normal is None else d_normal.contiguous()
        g = node.field.backward(d_sdf, d_rgb, d_normal, ctx.B)
        d_tfs = g["tfs"].reshape(ctx.tfs_shape)
        d_time = g["time_code"] if ctx.has_time else None
        return (None, None, None, None, None, None, d_tfs, g["pose_embed"], d_time,
                *g["iw"], *g["ib"], *g["rw"], *g["rb"])


def _check_gen(field, gen, what):
    if field.gen != gen:
        raise RuntimeError(
            f"hold_amd: {what}.backward() after the node's activation buffers were overwritten by a later forward "
            "(activations live in a per-node pool, not in the autograd graph): call backward() before the next "
            "forward of the same node, or use a separate HOLDNet instance for the interleaved evaluation")


class _EikonalFn(torch.autograd.Function):
    """g = d sdf / d x at free canonical points with a differentiable (second-order) backward to the weights:
    compute_gradient_samples + compute_gradient(create_graph=True), code/src/engine/volsdf_utils.py:6-48."""

    @staticmethod
    def forward(ctx, inet, xc, barf_w, *weights):
        iw, ib = weights[0:9], weights[9:18]
        fld = inet._field(xc.device, "eik")
        pk = inet._pack(fld.spec, iw, ib)
        g = fld.grad_points_forward(pk, xc, xc.shape[0], barf_w)
        ctx.fld, ctx.gen = fld, fld.gen
        return g[:, :3].clone()

    @staticmethod
    def backward(ctx, gbar):
        _check_gen(ctx.fld, ctx.gen, "ImplicitNet.gradient")
        g_iw, g_ib = ctx.fld.grad_points_backward(gbar.contiguous())
        return (None, None, None, *g_iw, *g_ib)


class _ImplicitFn(torch.autograd.Function):
    """ImplicitNet.forward at canonical points (shape_net.py:84-130) with its first-order backward."""

    @staticmethod
    def forward(ctx, inet, x, barf_w, *weights):
        iw, ib = weights[0:9], weights[9:18]
        fld = inet._field(x.device, "oc")
        pk = inet._pack(fld.spec, iw, ib)
        P = x.shape[0]
        xc = fld.pool.get("xc_in", P, 4)
        xc[:, :3] = x.detach()
        o = fld.sdf_feat_forward(pk, xc, P, barf_w)
        ctx.fld, ctx.gen, ctx.P = fld, fld.gen, P
        return torch.cat([o[:, 256:257], o[:, :256]], 1)

    @staticmethod
    def backward(ctx, d_out):
        _check_gen(ctx.fld, ctx.gen, "ImplicitNet.forward")
        fld, P = ctx.fld, ctx.P
        ob = fld.pool.get("oc_bar", P, 260)
        ob.zero_()
        ob[:, :256] = d_out[:, 1:]
        ob[:, 256] = d_out[:, 0]
        g_iw, g_ib, xbar = fld.sdf_feat_backward(ob)
        return (None, xbar[:, :3].clone(), None, *g_iw, *g_ib)


class _CompositeFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, S, n_rays, class_ids, want_w, betas, *args):
        n = len(class_ids)
        z, sdf, color, normal, beta = (args[i * n:(i + 1) * n] for i in range(5))
        dev = sdf[0].device
        betas = [float(b) for b in (beta if betas is None else betas)]  # host values of `beta` (kernel scalars)
        sdf = [s.contiguous() for s in sdf]
        color = [c if c.stride(-1) == 1 else c.contiguous() for c in color]
        normal = [c if c.stride(-1) == 1 else c.contiguous() for c in normal]
        d = K.make_composite_desc(S, n_rays, z, sdf, color, normal, class_ids, betas)
        out_node = [torch.empty(n_rays, 12, device=dev) for _ in range(n)]
        out_comp = torch.empty(n_rays, 12, device=dev)
        out_sem = torch.empty(n_rays, 4, device=dev)
        M = n * S - 2 * n + 1
        out_w = torch.empty(n_rays, M, device=dev) if want_w else None
        out_wn = [torch.empty(n_rays, S, device=dev) for _ in range(n)] if want_w else None
        K.composite_fwd(d, out_node, out_comp, out_sem, out_w, out_w_node=out_wn)
        ctx.save_for_backward(*z, *sdf, *color, *normal)
        ctx.meta = (S, n_rays, class_ids, betas, n)
        if out_w is None:
            out_w = torch.empty(0, device=dev)
            out_wn = [torch.empty(0, device=dev) for _ in range(n)]
        ctx.mark_non_differentiable(out_w, *out_wn)
        return (out_comp, out_sem, out_w, *out_node, *out_wn)

    @staticmethod
    def backward(ctx, d_comp, d_sem, d_w, *d_rest):
        S, n_rays, class_ids, betas, n = ctx.meta
        d_node = d_rest[:n]  # the per-node weights (d_rest[n:]) are non-differentiable outputs
        sv = ctx.saved_tensors
        z, sdf, color, normal = (sv[i * n:(i + 1) * n] for i in range(4))
        dev = sdf[0].device
        zeros12 = lambda t: torch.zeros(n_rays, 12, device=dev) if t is None else t.contiguous()
        d_comp = zeros12(d_comp)
        d_sem = torch.zeros(n_rays, 4, device=dev) if d_sem is None else d_sem.contiguous()
        d_node = [zeros12(t) for t in d_node]
        d = K.make_composite_desc(S, n_rays, z, sdf, color, normal, class_ids, betas)
        d_sdf = [torch.empty(n_rays, S, device=dev) for _ in range(n)]
        d_color = [torch.empty(n_rays * S, 3, device=dev) for _ in range(n)]
        d_normal = [torch.empty(n_rays * S, 3, device=dev) for _ in range(n)]
        d_beta = torch.zeros(3, device=dev)
        K.composite_bwd(d, d_node, d_comp, d_sem, d_sdf, d_color, d_normal, d_beta)
        d_sdf = [t.view(-1) for t in d_sdf]
        return (None, None, None, None, None, *([None] * n), *d_sdf, *d_color, *d_normal, *[d_beta[i] for i in range(n)])


class _BackgroundFn(torch.autograd.Function):
    """Background.bg_rendering (background.py:56-100) fwd + bwd on the kernels."""

    @staticmethod
    def forward(ctx, bg, ray_dirs, cam_loc, z_bg, rays_per_frame, latent, *weights):
        out = bg._fwd(ray_dirs, cam_loc, z_bg, rays_per_frame, latent.detach().contiguous(), weights)
        ctx.bg = bg
        ctx.B = latent.shape[0]
        return out

    @staticmethod
    def backward(ctx, d_out):
        g = ctx.bg._bwd(d_out.contiguous(), ctx.B)
        return (None, None, None, None, None, g["latent"], *g["iw"], *g["ib"], *g["rw"], *g["rb"])


# ------------------------------------------------------------------------------------------ nodes
class Node(nn.Module):
    """Node of code/src/model/renderables/node.py:12-109: networks, sampler, density, deformer, server, pose tables."""

    def __init__(self, node_id, kind, n_frames, sdf_bounding_sphere, sampler_opt, barf_s, barf_e, no_barf, params, server,
                 init="geometry", init_bias=0.6):
        super().__init__()
        self.node_id, self.kind, self.class_id = node_id, kind, CLASS_ID[node_id]
        self.spec = FieldSpec(kind)
        self.sdf_bounding_sphere = sdf_bounding_sphere
        cond = 45 if kind == "hand" else 0
        # `params` was constructed by the caller BEFORE the networks, as the reference does, so that the same torch
        # seed gives the same initial parameters; sub-modules are registered in the reference's order (node.py:27-42)
        self.implicit_network = ImplicitNet(3, 6, cond, True, "fourier" if kind == "hand" else "barf", barf_s, barf_e,
                                            no_barf, init=init, bias=init_bias)
        self.rendering_network = RenderingNet("pose", self.spec.rin_dim, [256] * 4, True, pose_dim=cond)
        self.ray_sampler = ErrorBoundSampler(sdf_bounding_sphere, inverse_sphere_bg=True, **sampler_opt)
        self.density = LaplaceDensity()
        self.server = server
        self.params = params
        self.field = None
        self._pk = None
        self._pk_key = None

    def _field(self, device):
        if self.field is None or self.field.device != device:
            self.field = NodeField(self.spec, device)
        return self.field

    def step_embedding(self):
        self.implicit_network.embedder_obj.step()

    def meshing_cano(self, pose=None):  # node.py:44-45
        return None

    def eikonal_grad(self, points):
        """grad_theta of the reference (volsdf_utils.compute_gradient_samples): d sdf / d x at canonical sample
        points [B, n, 3] -> [B, n, 3], differentiable w.r.t. the implicit network's parameters."""
        B, n, _ = points.shape
        return self.implicit_network.gradient(points.reshape(-1, 3)).view(B, n, 3)

    def query_oc(self, x):
        """hold_utils.query_oc (code/src/hold/hold_utils.py:61-65): canonical SDF at x [B,n,3] -> [B,n]."""
        return self.implicit_network(x, None)[:, :, 0]

    def _gradient_samples(self, sampler, centers, num, local_sigma, global_ratio):
        """compute_gradient_samples (volsdf_utils.py:19-48): randperm(V)[:num] centres (CPU generator, as the
        reference), one Gaussian copy each + global_ratio * num uniform box samples -> d sdf / d x with graph."""
        idx = _lib_h2d(torch.randperm(centers.shape[1])[:num], centers.device)
        sample = sampler.get_points(torch.index_select(centers, 1, idx), local_sigma=local_sigma,
                                    global_ratio=global_ratio)
        return sample, self.eikonal_grad(sample)

    def _weights(self):
        iw, ib = self.implicit_network.effective()
        rw, rb = self.rendering_network.effective()
        return (*iw, *ib, *rw, *rb)

    def render(self, input, ray_dirs, cam_loc, rays_per_frame, rng=None, z_override=None):
        """Node.forward (node.py:49-87): sample -> canonical sdf/feature -> colour/normal.  Returns factors."""
        dev = ray_dirs.device
        field = self._field(dev)
        training = self.training
        so, tfs, cond_pose, time_code = self.serve(input)
        dfm_const = self.deform_const(so)
        barf_w = self.implicit_network.embedder_obj.weights(dev)
        weights = self._weights()
        nb = self.spec.n_bones
        N = ray_dirs.shape[0]
        # one re-layout of the weights for sampler + field + backward, kept until a parameter changes (every chunk of a
        # step, and every eval forward between optimiser steps, reuses it)
        key = _pack_key((self.implicit_network, self.rendering_network), training)
        if self._pk is None or self._pk_key != key:
            with torch.no_grad():  # the trunk's part is shared with ImplicitNet.forward / gradient (one re-layout per step)
                trunk = self.implicit_network._pack(self.spec, weights[0:9], weights[9:18])
                self._pk = pack_weights(self.spec, weights[0:9], weights[9:18], weights[18:23], weights[23:28], training,
                                        trunk=trunk)
            self._pk_key = key
        # ---- sampler (no grad; sampler toggles net.eval()/train() in the reference, a no-op for these nets) ----
        if z_override is None:
            with torch.no_grad():
                pk = self._pk
                dfm = dict(dfm_const)
                dfm["tfs"] = tfs.detach().reshape(-1, nb, 16).contiguous()

                def sdf_query(x, P, out):
                    field.sdf_only(pk, x, P, P // tfs.shape[0], dfm, barf_w, out)

                z_vals = self.ray_sampler.sample_z(sdf_query, ray_dirs, cam_loc, self.density.beta_host(),
                                                   training, rng)
        else:
            z_vals = z_override.contiguous()
        S = z_vals.shape[1]
        x = field.pool.get("x_pts", N * S, 4)
        K.ray_points(cam_loc, ray_dirs, z_vals, S, x)
        if self.kind == "hand":
            pose_embed = self.rendering_network.lin_pose(cond_pose)
        else:
            pose_embed = torch.zeros(tfs.shape[0], 8, device=dev)
        sdf, rgb, normal, xc = _FieldFn.apply(self, x, rays_per_frame * S, dfm_const, barf_w, training, tfs, pose_embed,
                                              time_code, *weights)
        return dict(z_vals=z_vals, sdf=sdf, color=rgb, normal=normal, canonical_pts=xc, server=so, tfs=tfs)


class MANONode(Node):
    """code/src/model/renderables/mano_node.py:17-151."""

    def __init__(self, node_id, betas, n_frames, sdf_bounding_sphere, sampler_opt, mano_model, barf_s, barf_e,
                 no_barf, init="geometry", init_bias=0.6):
        server = MANOServer(betas, node_id == "right", mano_model)
        params = GenericParams(n_frames, {"betas": 10, "global_orient": 3, "transl": 3, "pose": 45}, node_id)
        super().__init__(node_id, "hand", n_frames, sdf_bounding_sphere, sampler_opt, barf_s, barf_e, no_barf, params,
                         server, init, init_bias)
        self.is_rhand = node_id == "right"
        from .deformer import MANODeformer
        self.deformer = MANODeformer(max_dist=0.1, K=15, betas=betas, is_rhand=self.is_rhand, server=self.server)
        self.register_buffer("mesh_f_cano", torch.as_tensor(self.server.faces.astype(np.int64)), persistent=False)
        # loss-target mesh: the sealed, once-subdivided canonical MANO, spawned every 200 steps (hold_net.py:161-166)
        self.mesh_v_cano_div = None
        self.mesh_f_cano_div = None
        self.canonical_mesh = None
        self.pt_sampler = PointInSpace(global_sigma_xyz=[0.15, 0.06, 0.12])  # hold_utils.py:58

    # canonical vertices of the KNN deformer (mano/deformer.py:20-32) = the server's canonical pose output
    cano_verts = property(lambda self: self.server.verts_c[0])
    mesh_v_cano = property(lambda self: self.server.verts_c)

    def sample_eikonal_points(self, batch_size, num=256, local_sigma=0.008, global_ratio=0.20):
        """the sample points of compute_gradient_samples around random canonical MANO vertices."""
        v = self.cano_verts[None].expand(batch_size, -1, -1)
        idx = torch.randperm(v.shape[1])[:num].to(v.device)
        return self.pt_sampler.get_points(torch.index_select(v, 1, idx), local_sigma=local_sigma,
                                          global_ratio=global_ratio)

    def spawn_cano_mano(self, sample_dict_h):
        """mano_node.py:126-135: seal the pose-corrected canonical vertices (first frame of the batch) and Loop-
        subdivide once -> 3 110 vertices / 6 216 faces."""
        so = sample_dict_h.get("output", sample_dict_h)
        v, f = seal_mano_mesh(so["v_posed"].detach(), self.mesh_f_cano, self.is_rhand)
        self.mesh_v_cano_div, self.mesh_f_cano_div = subdivide_loop(v[0].float(), f)
        self._mesh_index = MeshIndex(self.mesh_v_cano_div, self.mesh_f_cano_div, 0.01)  # threshold of hold_utils.py:222

    def meshing_cano(self, pose=None):
        """mano_node.py:137-151: canonical mesh of the learnt SDF inside MANO's canonical bounding box."""
        from .meshing import generate_mesh
        v_min_max = np.array([[-0.0814, -0.0280, -0.0742], [0.1171, 0.0349, 0.0971]])
        dev = self.mesh_f_cano.device
        return generate_mesh(lambda x: {"sdf": self.implicit_network.sdf(x)}, v_min_max, point_batch=10000, res_up=1,
                             res_init=64, device=dev)

    def loss_targets(self, out, fac, B, n_pix, frame_terms=True):
        """prepare_loss_targets_hand (code/src/hold/hold_utils.py:186-240)."""
        if self.mesh_v_cano_div is None:
            return
        nid = self.node_id
        mesh_v = self.mesh_v_cano_div[None]
        mesh_f = self.mesh_f_cano_div
        if frame_terms:
            samples = sample_on_barycentric_mesh(mesh_v.expand(B, -1, -1), mesh_f, num_samples=256)
            samples = self.pt_sampler.get_points(samples, local_sigma=0.008, global_ratio=0.20)
            out[f"{nid}.pts2mano_sdf_cano"] = compute_mano_cano_sdf(mesh_v[0], mesh_f, samples)
            out[f"{nid}.pred_sdf"] = self.query_oc(samples)
        out[f"{nid}.index_off_surface"] = self._mesh_index.off_surface(fac["canonical_pts"], B * n_pix)
        if frame_terms:
            verts_c = self.cano_verts[None].expand(B, -1, -1)
            esamp, g = self._gradient_samples(self.pt_sampler, verts_c, 256, 0.008, 0.20)
            out[f"{nid}.grad_theta"] = g
            self._last_targets = dict(mano_cano_samples=samples, eikonal_samples=esamp)

    def serve(self, input):
        nid = self.node_id
        full_pose = input[f"{nid}.full_pose"]
        so = self.server(input[f"{nid}.params"][:, 0], input[f"{nid}.transl"], full_pose, input[f"{nid}.betas"])
        cond = full_pose[:, 3:] / np.pi
        if self.training and input.get("current_epoch", 0) < 20:
            cond = full_pose[:, 3:] * 0.0  # mano_node.py:82-85
        return so, so["tfs"], cond, None

    def deform_const(self, so):
        return dict(verts=so["verts"].detach().contiguous(), skin_w=self.server.human_layer.lbs_weights.contiguous(),
                    verts_c=self.cano_verts.contiguous())


class ObjectNode(Node):
    """code/src/model/renderables/object_node.py:17-132."""

    def __init__(self, node_id, n_frames, sdf_bounding_sphere, sampler_opt, entity, barf_s, barf_e, no_barf,
                 init="geometry", init_bias=0.6):
        params = GenericParams(n_frames, {"global_orient": 3, "transl": 3}, node_id)
        server = ObjectServer(entity)
        super().__init__(node_id, "object", n_frames, sdf_bounding_sphere, sampler_opt, barf_s, barf_e, no_barf, params,
                         server, init, init_bias)
        from .deformer import ObjectDeformer
        self.deformer = ObjectDeformer()
        self.frame_latent_encoder = nn.Embedding(n_frames, 32)
        self.is_test = False
        self.mesh_o = None
        self.mesh_vo_cano = None
        self.mesh_fo_cano = None
        v3d = np.asarray(entity["pts.cano"], dtype=np.float32)
        self.v_min_max = np.array([v3d.min(axis=0), v3d.max(axis=0)]) * 2.0  # object_node.py:49-50

    def meshing_cano(self, pose=None):
        """object_node.py:112-121."""
        from .meshing import generate_mesh
        dev = self.frame_latent_encoder.weight.device
        mesh = generate_mesh(lambda x: {"sdf": self.implicit_network.sdf(x)}, self.v_min_max, point_batch=10000,
                             res_up=2, device=dev)
        self.update_cano(mesh)
        return mesh

    def update_cano(self, mesh_canonical):
        """object_node.py:123-132 (``mesh_o``, kaolin's face-vertex tensor, is kept as the face-gathered vertices)."""
        dev = self.frame_latent_encoder.weight.device
        if mesh_canonical is None or len(np.asarray(mesh_canonical.vertices)) == 0 or len(np.asarray(mesh_canonical.faces)) == 0:
            return  # no zero crossing in the box (the reference's generate_mesh returns None): keep the previous targets
        self.mesh_vo_cano = torch.as_tensor(np.asarray(mesh_canonical.vertices)[None], device=dev).float()
        self.mesh_fo_cano = torch.as_tensor(np.asarray(mesh_canonical.faces).astype(np.int64), device=dev)
        self.mesh_o = self.mesh_vo_cano[:, self.mesh_fo_cano]
        self._mesh_index = MeshIndex(self.mesh_vo_cano[0], self.mesh_fo_cano, 0.05)  # threshold of hold_utils.py:166

    def loss_targets(self, out, fac, B, n_pix, frame_terms=True):
        """prepare_loss_targets_object (code/src/hold/hold_utils.py:149-183)."""
        if self.mesh_o is None:
            return
        nid = self.node_id
        out[f"{nid}.index_off_surface"] = self._mesh_index.off_surface(fac["canonical_pts"], B * n_pix)
        if frame_terms:
            xyz = self.mesh_vo_cano[0].abs().max(dim=0).values * 1.1
            sampler = PointInSpace(global_sigma_xyz=xyz)
            esamp, g = self._gradient_samples(sampler, self.mesh_vo_cano.expand(B, -1, -1), 256, 0.03, 0.20)
            out[f"{nid}.grad_theta"] = g
            self._last_targets = dict(eikonal_samples=esamp)

    def serve(self, input):
        nid = self.node_id
        o = self.server.object_model(rot=input[f"{nid}.global_orient"], trans=input[f"{nid}.transl"],
                                     scene_scale=input[f"{nid}.params"][:, 0], want_verts=False)
        so = {"obj_tfs": o["T"][:, None]}
        return so, so["obj_tfs"], None, self.frame_latent_encoder(input["idx"])

    def deform_const(self, so):
        return {}


# ------------------------------------------------------------------------------------------ background
class Background(nn.Module):
    def __init__(self, num_frames, sdf_bounding_sphere):
        super().__init__()
        self.bg_implicit_network = ImplicitNet(4, 10, 32, False)
        self.bg_rendering_network = RenderingNet("nerf_frame_encoding", 27 + 32 + FEAT, [128], False)
        self.frame_latent_encoder = nn.Embedding(num_frames, 32)
        self.sdf_bounding_sphere = sdf_bounding_sphere
        self.inverse_sphere_sampler = UniformSampler(1.0, 0.0, 32, False, far=1.0)
        self.pool = None
        self.E, self.K0, self.skip_out = 84, 116, 172
        self.Kr = pad4(27 + 32 + FEAT)  # 316

    def step_embedding(self):
        pass

    def render(self, ray_dirs, cam_loc, z_bg, rays_per_frame, idx):
        iw, ib = self.bg_implicit_network.effective()
        rw, rb = self.bg_rendering_network.effective()
        return _BackgroundFn.apply(self, ray_dirs, cam_loc, z_bg, rays_per_frame, self.frame_latent_encoder(idx),
                                   *iw, *ib, *rw, *rb)

    def _fwd(self, ray_dirs, cam_loc, z_bg, rays_per_frame, latent, weights):
        iw, ib, rw, rb = weights[0:9], weights[9:18], weights[18:20], weights[20:22]
        dev = ray_dirs.device
        if self.pool is None or self.pool.device != dev:
            self.pool = Pool(dev)
        pool = self.pool
        N, S = z_bg.shape
        P = N * S
        ppf = rays_per_frame * S
        zf = torch.flip(z_bg, dims=[-1]).contiguous()
        pts = pool.get("pts", P, 4)
        K.bg_points(cam_loc, ray_dirs, zf, S, self.sdf_bounding_sphere, pts)
        in0 = pool.get("in0", P, self.K0)
        h = [pool.get(f"h{l}", P, 256) for l in range(8)]
        K.embed_fwd(pts, 4, 10, P, in0, out2=h[3][:, self.skip_out:], cond=latent, pts_per_frame=ppf)
        W = [w.contiguous() for w in iw]
        W[4] = (iw[4] / math.sqrt(2)).contiguous()
        W[8] = torch.cat([iw[8][1:], iw[8][:1]], 0).contiguous()
        b = [t.contiguous() for t in ib]
        b[8] = torch.cat([ib[8][1:], ib[8][:1]]).contiguous()
        G.gemm_nt(in0, W[0], h[0], bias=b[0], epi=G.EPI_SOFTPLUS, K=self.K0)
        G.gemm_nt(h[0], W[1], h[1], bias=b[1], epi=G.EPI_SOFTPLUS)
        G.gemm_nt(h[1], W[2], h[2], bias=b[2], epi=G.EPI_SOFTPLUS)
        G.gemm_nt(h[2], W[3], h[3][:, :self.skip_out], bias=b[3], epi=G.EPI_SOFTPLUS, N=self.skip_out)
        for l in range(4, 8):
            G.gemm_nt(h[l - 1], W[l], h[l], bias=b[l], epi=G.EPI_SOFTPLUS)
        rin = pool.get("rin", P, self.Kr)
        sdf = pool.get("sdf", P, 1)
        # lin8 = 256 feature rows as one full-tile GEMM + the sdf row as a row dot (as the foreground nets do: N = 257 costs a
        # second 256-wide column tile for the one extra output)
        G.gemm_nt(h[7], W[8], rin[:, 59:59 + FEAT], bias=b[8], N=256)
        K.rowdot(h[7], iw[8][0].contiguous(), 256, ib[8][:1].contiguous(), P, sdf)
        dirs = pool.get("dirs", P, 4)
        K.frame_bcast(ray_dirs, P, S, dirs, 0)
        K.embed_fwd(dirs, 3, 4, P, rin, cond=latent, pts_per_frame=ppf)
        R0 = torch.zeros(128, self.Kr, device=dev)
        R0[:, :315] = rw[0]
        R1 = rw[1].contiguous()
        r0 = pool.get("r0", P, 128)
        G.gemm_nt(rin, R0, r0, bias=rb[0].contiguous(), epi=G.EPI_RELU, K=self.Kr)
        rgb = pool.get("rgb", P, 4)
        G.head3_fwd(r0, R1, rb[1].contiguous(), rgb, K=128)
        out = torch.empty(N, 3, device=dev)
        K.bg_composite_fwd(zf, sdf, rgb, S, N, out)
        self.saved = dict(P=P, N=N, S=S, ppf=ppf, zf=zf, in0=in0, h=h, rin=rin, sdf=sdf, r0=r0, rgb=rgb, W=W, R0=R0,
                          R1=R1)
        return out

    def _bwd(self, d_out, B):
        sv, pool = self.saved, self.pool
        P, N, S, ppf = sv["P"], sv["N"], sv["S"], sv["ppf"]
        h, rin, W = sv["h"], sv["rin"], sv["W"]
        dev = d_out.device
        d_sdf = pool.get("d_sdf", P, 1)
        d_rgb = pool.get("d_rgb", P, 3)
        K.bg_composite_bwd(sv["zf"], sv["sdf"], sv["rgb"], S, N, d_out, d_sdf, d_rgb)
        dy = pool.get("dy", P, 4)
        sg = sv["rgb"][:, :3]
        dy[:, :3] = d_rgb * sg * (1 - sg)
        dR1, db4 = torch.zeros(3, 128, device=dev), torch.zeros(4, device=dev)
        rr0 = pool.get("rr0", P, 128)
        G.head3_bwd(dy, sv["r0"], sv["R1"], rr0, dR1, db4, K=128)
        dR1b = db4[:3]
        dR0, dR0b = torch.zeros(128, self.Kr, device=dev), torch.zeros(128, device=dev)
        G.wgrad(rr0, rin, dR0, dR0b, K=self.Kr)
        d_rin = pool.get("d_rin", P, self.Kr)
        G.gemm_nt(rr0, sv["R0"].t().contiguous(), d_rin, N=self.Kr)
        d_lat = torch.zeros(B, 32, device=dev)
        K.frame_colsum(d_rin, 27, 32, P, ppf, d_lat)
        # implicit net, first-order sweep only (no normals in the background)
        ob = pool.get("out_bar", P, 260)
        K.copy_cols(d_rin[:, 59:59 + FEAT], ob, FEAT, P)
        K.copy_cols(d_sdf, ob[:, 256:257], 1, P)
        dW, dWb = zeros_like_many(W, [w[:, 0] for w in W])
        WT = []
        for l in range(9):
            n, k = W[l].shape
            wt = torch.zeros(k, pad4(n), device=dev)
            wt[:, :n] = W[l].t()
            WT.append(wt)
        G.wgrad(ob, h[7], dW[8], dWb[8], N=257)
        rb_ = [pool.get(f"rb{i}", P, 256) for i in range(2)]
        G.gemm_nt(ob, WT[8], rb_[0], epi=G.EPI_MUL_DSP, aux1=h[7], K=260)
        cur = rb_[0]
        so = self.skip_out
        if config.x6() and _field.USE_R6_BWD:
            # the seven 256-wide layers as ONE register-resident descending sweep (csrc/rchain.hip, skip width 172) instead
            # of seven hold_gemm_nt launches with the MUL_DSP epilogue; every r_l stays for the weight gradients
            # (advisor r4: the pack -- limb split + permutation of 7 x 256 x 256 -- was rebuilt on EVERY backward call, i.e. per ray
            # chunk; the weights change once per optimiser step: cached on the weights epoch and the parameters' versions)
            wkey = _pack_key([self.bg_implicit_network], True)
            if getattr(self, "_wr6_cache", (None, None))[0] != wkey:
                stacked = torch.stack([F.pad(W[l], (0, 0, 0, 256 - W[l].shape[0])) for l in range(1, 8)])
                MT = stacked.transpose(1, 2).flip(0).contiguous()  # layer j = W_{7-j}^T
                h3 = None
                if config.h3() and _field.USE_H3_BWD:  # the same sweep in two fp16 limbs (hold_chain_h3, skip width 172)
                    pk3, sw3 = _field.pack_h3_stack(MT)
                    h3 = dict(wpack_h3=pk3, c3=(1.0 / sw3).contiguous())
                self._wr6_cache = (wkey, _field.pack_r6_stack(MT), h3)
            wr6 = self._wr6_cache[1]
            r = [pool.get(f"rs{l}", P, 256) for l in range(7)] + [cur]
            K.chain(K.CHAIN_DSP, P, cur, None, 7, 32, skip_layer=3, aux1=[h[l - 1] for l in range(7, 0, -1)],
                    out=[r[l - 1] for l in range(7, 0, -1)], wpack_r6=wr6, skip_out=so, **(self._wr6_cache[2] or {}))
            grp = G.WgradGroup() if _field.USE_WGRAD_GROUP else None  # every r_l has a buffer of its own: one launch for the 7
            wg = G.wgrad if grp is None else grp.add
            for l in range(7, 0, -1):
                if l == 3:
                    wg(r[3], h[2], dW[3], dWb[3], N=so)
                else:
                    wg(r[l], h[l - 1], dW[l], dWb[l])
            if grp is not None:
                grp.flush()
            cur = r[0]
        else:
            for l in range(7, 0, -1):
                nxt = rb_[1] if cur is rb_[0] else rb_[0]
                if l == 4:
                    G.wgrad(cur, h[3], dW[4], dWb[4])
                    G.gemm_nt(cur, WT[4], nxt[:, :so], epi=G.EPI_MUL_DSP, aux1=h[3], N=so)
                elif l == 3:
                    G.wgrad(cur, h[2], dW[3], dWb[3], N=so)
                    G.gemm_nt(cur, WT[3], nxt, epi=G.EPI_MUL_DSP, aux1=h[2], K=so)
                else:
                    G.wgrad(cur, h[l - 1], dW[l], dWb[l])
                    G.gemm_nt(cur, WT[l], nxt, epi=G.EPI_MUL_DSP, aux1=h[l - 1])
                cur = nxt
        G.wgrad(cur, sv["in0"], dW[0], dWb[0], K=self.K0)
        d_in0 = pool.get("d_in0", P, self.K0)
        G.gemm_nt(cur, WT[0], d_in0, N=self.K0)
        K.frame_colsum(d_in0, self.E, 32, P, ppf, d_lat)
        g_iw = dW[:4] + [dW[4] / math.sqrt(2)] + dW[5:8] + [torch.cat([dW[8][256:257], dW[8][:256]], 0)]
        g_ib = dWb[:8] + [torch.cat([dWb[8][256:257], dWb[8][:256]])]
        return dict(latent=d_lat, iw=g_iw, ib=g_ib, rw=[dR0[:, :315], dR1], rb=[dR0b, dR1b])


# ------------------------------------------------------------------------------------------ HOLDNet
def get_camera_params(uv, pose, intrinsics):
    """code/src/datasets/utils.py:256-282 (pose-matrix branch) on hold_raygen: -> (ray_dirs [B,P,3], cam_loc [B,3])."""
    B, Pn, _ = uv.shape
    dirs, _ = K.raygen(uv, pose, intrinsics)
    return dirs.view(B, Pn, 3), pose[:, :3, 3]


DEFAULT_SAMPLER = dict(near=0.0, N_samples=64, N_samples_eval=128, N_samples_extra=32, eps=0.1, beta_iters=10,
                       max_total_iters=5, N_samples_inverse_sphere=32, add_tiny=1e-6)


class HOLDNet(nn.Module):
    """HOLDNet of code/src/hold/hold_net.py:23-179 with the file-backed inputs (MANO pickles, data.npy entities) passed
    explicitly; ``hold_amd.ReferenceHOLDNet`` wraps it in the reference's ``(opt, betas_r, betas_l, num_frames, args)``
    constructor.  ``forward(input) -> xdict`` with the reference's keys, including the training-only loss targets
    ``<node>.{index_off_surface, grad_theta, pts2mano_sdf_cano, pred_sdf}`` (prepare_loss_targets, :154-179)."""

    def __init__(self, scene_bounding_sphere, betas_r, betas_l, num_frames, entities, mano_models, sampler_opt=None,
                 barf_s=1000, barf_e=10000, no_barf=False, init="geometry", init_bias=0.6, load_pose_tables=True):
        super().__init__()
        self.sdf_bounding_sphere = float(scene_bounding_sphere)
        self.threshold = 0.05
        so = dict(DEFAULT_SAMPLER if sampler_opt is None else sampler_opt)
        nodes = {}
        if betas_r is not None:
            nodes["right"] = MANONode("right", betas_r, num_frames, self.sdf_bounding_sphere, so, mano_models["right"],
                                      barf_s, barf_e, no_barf, init, init_bias)
        if betas_l is not None:
            nodes["left"] = MANONode("left", betas_l, num_frames, self.sdf_bounding_sphere, so, mano_models["left"],
                                     barf_s, barf_e, no_barf, init, init_bias)
        nodes["object"] = ObjectNode("object", num_frames, self.sdf_bounding_sphere, so, entities["object"], barf_s,
                                     barf_e, no_barf, init, init_bias)
        self.nodes = nn.ModuleDict(nodes)
        self.background = Background(num_frames, self.sdf_bounding_sphere)
        if load_pose_tables:  # params.load_params(args.case) of mano_node.py:46 / object_node.py:33
            for nid, node in self.nodes.items():
                ent = entities.get(nid, {})
                if ("hand_poses" in ent) or ("object_poses" in ent):
                    node.params.load_entity(ent)
        self.auto_step_embedding = True  # train_step() steps the BARF counter once per optimiser step instead
        # weight packs are cached per node and keyed on torch's version counters; writes those counters do not see
        # (load_state_dict copies through .data on some paths, .to() / .float() re-create the storages) bump the epoch
        self.register_load_state_dict_post_hook(lambda module, incompatible: config.bump_weights_epoch())

    def _apply(self, fn, *args, **kwargs):
        out = super()._apply(fn, *args, **kwargs)
        config.bump_weights_epoch()
        return out

    def init_network(self, shape_init=""):
        """hold_net.py:136-152: optionally start the hand / object SDF nets from a pre-trained checkpoint."""
        if not shape_init:
            return
        sd = torch.load(f"./saved_models/{shape_init}/checkpoints/last.ckpt", map_location="cpu")["state_dict"]
        sd = {k.replace("model.", ""): v for k, v in sd.items()
              if "implicit_network" in k and "bg_implicit_network." not in k and ".embedder_obj." not in k}
        self.load_state_dict(sd, strict=False)

    def step_embedding(self):
        for node in self.nodes.values():
            node.step_embedding()
        self.background.step_embedding()

    def prepare_loss_targets(self, out, fac, step, B, n_pix, frame_terms=True):
        """hold_net.py:154-179 + hold_utils.prepare_loss_targets_{hand,object}."""
        if step % 200 == 0 and step > 0:
            for nid, node in self.nodes.items():
                if nid in ("right", "left"):
                    node.spawn_cano_mano(fac[nid]["server"])
        for nid, node in self.nodes.items():
            node.loss_targets(out, fac[nid], B, n_pix, frame_terms)

    def forward(self, input, rng=None, z_override=None):
        if not torch.cuda.is_available():
            raise RuntimeError("hold_amd.HOLDNet needs an MI355X: the hot path has no CPU / eager fallback")
        training = self.training
        XD = output_class()
        with torch.enable_grad() if training else torch.no_grad():
            B, Pn, _ = input["uv"].shape
            ray_dirs, cam_loc = K.raygen(input["uv"], input["extrinsics"], input["intrinsics"])  # [N,3] each
            N = B * Pn
            out = {}
            if training:
                out["epoch"], out["step"] = input["current_epoch"], input["global_step"]
            fac = {}
            for nid, node in self.nodes.items():
                fac[nid] = node.render(input, ray_dirs, cam_loc, Pn, None if rng is None else rng.get(nid),
                                       None if z_override is None else z_override[nid])
            if training:
                self.prepare_loss_targets(out, fac, int(out["step"]), B, Pn, input.get("hold_amd.frame_terms", True))
            ids = list(fac.keys())
            S = fac[ids[0]]["z_vals"].shape[1]
            args = ([fac[i]["z_vals"] for i in ids] + [fac[i]["sdf"] for i in ids] + [fac[i]["color"] for i in ids] +
                    [fac[i]["normal"] for i in ids] + [self.nodes[i].density.get_beta() for i in ids])
            res = _CompositeFn.apply(S, N, [self.nodes[i].class_id for i in ids], True,
                                     [self.nodes[i].density.beta_host() for i in ids], *args)
            comp, sem, w = res[0], res[1], res[2]

            def unpack(o, prefix, cls=None):
                d = {f"{prefix}fg_rgb": o[:, 0:3], f"{prefix}mask_prob": torch.clamp(o[:, 3:4], 0, 1),
                     f"{prefix}normal": o[:, 4:7], f"{prefix}depth": o[:, 7:8], f"{prefix}bg_weights": o[:, 8]}
                if cls is not None:
                    s_ = torch.zeros(N, 4, device=o.device)
                    s_[:, cls] = o[:, 3]
                    d[f"{prefix}fg_semantics"] = s_
                if not training:
                    d[f"{prefix}fg_rgb.vis"] = o[:, 0:3] + o[:, 8:9]
                return d

            out.update(unpack(comp, ""))
            out["fg_semantics"] = sem
            out["fg_weights"] = w
            for k, i in enumerate(ids):
                out.update(unpack(res[3 + k], f"{i}.", self.nodes[i].class_id))
                out[f"{i}.fg_weights"] = res[3 + len(ids) + k]
                out[f"{i}.z_vals"] = fac[i]["z_vals"]
            t_bg = None if rng is None else rng.get("bg_t")
            z_bg = self.background.inverse_sphere_sampler.inverse_sample(ray_dirs, cam_loc, training,
                                                                         self.sdf_bounding_sphere, t_bg)
            out["bg_z_vals"], out["ray_dirs"], out["cam_loc"], out["index"] = z_bg, ray_dirs, cam_loc, input["idx"]
            bg_only = self.background.render(ray_dirs, cam_loc, z_bg, Pn, input["idx"])
            bgw = out["bg_weights"].unsqueeze(-1)
            out["rgb"] = out["fg_rgb"] + bgw * bg_only
            bg_sem = torch.zeros(N, 4, device=bg_only.device)
            bg_sem[:, 0] = 1.0
            out["semantics"] = out["fg_semantics"] + bgw * bg_sem
            if not training:
                out["bg_rgb_only"] = bg_only
                out["instance_map"] = torch.argmax(out["semantics"], dim=1)
            self._last_factors = fac
            if training and self.auto_step_embedding:
                self.step_embedding()
        return XD(out)
