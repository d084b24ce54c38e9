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

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import gemm as G
from . import kernels as K
from .field import FEAT, FieldSpec, NodeField, Pool, pack_weights, pad4
from .mano import MANOServer, ObjectServer
from .sampler import ErrorBoundSampler, UniformSampler

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
        self.populate(self.alphas[int(self.alpha_iter)])

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
        self.alpha_iter = torch.tensor(min(int(self.alpha_iter) + 1, int(self.alpha_max_iter) - 1),
                                       device=self.alpha_iter.device)
        self.populate(self.alphas[int(self.alpha_iter)])

    def eval(self):  # embedders.py:124-125
        self.no_barf = True

    def weights(self, device):
        if self.no_barf:
            return None
        return self.barf_weights.to(device=device, dtype=torch.float32).contiguous()


def _lin(inn, out, weight_norm):
    l = nn.Linear(inn, out)
    return nn.utils.weight_norm(l) if weight_norm else l


def _eff(lin):
    """effective weight of a (possibly weight-normed) Linear, differentiable w.r.t. weight_g / weight_v."""
    if hasattr(lin, "weight_g"):
        v, g = lin.weight_v, lin.weight_g
        return v * (g / v.norm(dim=1, keepdim=True))
    return lin.weight


class ImplicitNet(nn.Module):
    """parameter container with the reference's names/shapes (shape_net.py:9-82)."""

    def __init__(self, d_in, multires, cond_dim, weight_norm, embedding="fourier", barf_s=1000, barf_e=10000,
                 no_barf=False):
        super().__init__()
        if embedding == "barf":
            self.embedder_obj = BarfEmbedder(d_in, multires, barf_s, barf_e, no_barf)
        else:
            self.embedder_obj = Embedder(d_in, multires)
        e = self.embedder_obj.out_dim
        self.d_in, self.multires, self.cond_dim, self.E = d_in, multires, cond_dim, e
        dims = [e] + [256] * 8 + [1 + FEAT]
        self.num_layers = len(dims)
        for l in range(9):
            out = dims[l + 1] - dims[0] if (l + 1) == 4 else dims[l + 1]
            inn = dims[l] + (cond_dim if l == 0 else 0)
            setattr(self, f"lin{l}", _lin(inn, out, weight_norm))

    def effective(self):
        lins = [getattr(self, f"lin{l}") for l in range(9)]
        return [_eff(l) for l in lins], [l.bias for l in lins]


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
        return [_eff(l) for l in lins], [l.bias for l in lins]


class LaplaceDensity(nn.Module):
    def __init__(self, beta=0.1, beta_min=1e-4):
        super().__init__()
        self.beta = nn.Parameter(torch.tensor(beta))
        self.beta_min = beta_min

    def get_beta(self):
        return self.beta.abs() + self.beta_min


class GenericParams(nn.Module):
    """per-frame pose tables (code/src/model/generic/params.py:6-62)."""

    def __init__(self, num_frames, params_dim, node_id):
        super().__init__()
        self.num_frames, self.params_dim, self.node_id = num_frames, params_dim, node_id
        self.param_names = list(params_dim.keys())
        for name, dim in params_dim.items():
            emb = nn.Embedding(1 if name == "betas" else num_frames, dim)
            emb.weight.data.fill_(0)
            emb.weight.requires_grad = False
            setattr(self, name, emb)

    def forward(self, frame_ids):
        out = {}
        for name in self.param_names:
            ids = torch.zeros_like(frame_ids) if name == "betas" else frame_ids
            out[f"{self.node_id}.{name}"] = getattr(self, name)(ids)
        if "pose" in self.param_names:
            out[f"{self.node_id}.full_pose"] = torch.cat(
                (out[f"{self.node_id}.global_orient"], out[f"{self.node_id}.pose"]), dim=1)
        return out

    def defrost(self, keys=None):
        for n in (keys or self.param_names):
            getattr(self, n).weight.requires_grad = True

    def freeze(self, keys=None):
        for n in (keys or self.param_names):
            getattr(self, n).weight.requires_grad = False


# ------------------------------------------------------------------------------------------ autograd glue
class _FieldFn(torch.autograd.Function):
    """sdf / colour / canonical normal of one node at given deformed-space points (fwd + hand-derived bwd)."""

    @staticmethod
    def forward(ctx, node, x, ppf, dfm_const, barf_w, training, tfs, pose_embed, time_code, *weights):
        iw, ib, rw, rb = weights[0:9], weights[9:18], weights[18:23], weights[23:28]
        pk = pack_weights(node.spec, iw, ib, rw, rb, need_bwd=training)
        P = x.shape[0]
        nb = node.spec.n_bones
        dfm = dict(dfm_const)
        dfm["tfs"] = tfs.detach().reshape(-1, nb, 16).contiguous()
        out = node.field.forward(pk, x, P, ppf, dfm, barf_w, pose_embed.detach().contiguous(),
                                 None if time_code is None else time_code.detach().contiguous(), training=training)
        ctx.node = node
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
        P = node.field.saved["P"]
        dev = node.field.device
        d_sdf = torch.zeros(P, device=dev) if d_sdf is None else d_sdf.contiguous()
        d_rgb = torch.zeros(P, 3, device=dev) if d_rgb is None else d_rgb.contiguous()
        d_normal = None if d_normal is None else d_normal.contiguous()
        g = node.field.backward(d_sdf, d_rgb, d_normal, ctx.B)
        d_tfs = g["tfs"].reshape(ctx.tfs_shape)
        d_time = g["time_code"] if ctx.has_time else None
        return (None, None, None, None, None, None, d_tfs, g["pose_embed"], d_time,
                *g["iw"], *g["ib"], *g["rw"], *g["rb"])


class _EikonalFn(torch.autograd.Function):
    """g = d sdf / d x at free canonical points with a differentiable (second-order) backward to the weights:
    compute_gradient_samples + compute_gradient(create_graph=True), code/src/engine/volsdf_utils.py:6-48."""

    @staticmethod
    def forward(ctx, node, xc, barf_w, *weights):
        iw, ib = weights[0:9], weights[9:18]
        rw, rb = weights[18:23], weights[23:28]
        pk = pack_weights(node.spec, iw, ib, rw, rb, need_bwd=True)
        fld = node._eik_field(xc.device)
        g = fld.grad_points_forward(pk, xc, xc.shape[0], barf_w)
        ctx.node = node
        return g[:, :3].clone()

    @staticmethod
    def backward(ctx, gbar):
        g_iw, g_ib = ctx.node._eik_field(gbar.device).grad_points_backward(gbar.contiguous())
        return (None, None, None, *g_iw, *g_ib, *([None] * 10))


class _CompositeFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, S, n_rays, class_ids, want_w, *args):
        n = len(class_ids)
        z, sdf, color, normal, beta = (args[i * n:(i + 1) * n] for i in range(5))
        dev = sdf[0].device
        betas = [float(b) for b in beta]
        sdf = [s.contiguous() for s in sdf]
        color = [c if c.stride(-1) == 1 else c.contiguous() for c in color]
        normal = [c if c.stride(-1) == 1 else c.contiguous() for c in normal]
        d = K.make_composite_desc(S, n_rays, z, sdf, color, normal, class_ids, betas)
        out_node = [torch.empty(n_rays, 12, device=dev) for _ in range(n)]
        out_comp = torch.empty(n_rays, 12, device=dev)
        out_sem = torch.empty(n_rays, 4, device=dev)
        M = n * S - 2 * n + 1
        out_w = torch.empty(n_rays, M, device=dev) if want_w else None
        K.composite_fwd(d, out_node, out_comp, out_sem, out_w)
        ctx.save_for_backward(*z, *sdf, *color, *normal)
        ctx.meta = (S, n_rays, class_ids, betas, n)
        if out_w is None:
            out_w = torch.empty(0, device=dev)
        ctx.mark_non_differentiable(out_w)
        return (out_comp, out_sem, out_w, *out_node)

    @staticmethod
    def backward(ctx, d_comp, d_sem, d_w, *d_node):
        S, n_rays, class_ids, betas, n = ctx.meta
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
        return (None, None, None, None, *([None] * n), *d_sdf, *d_color, *d_normal, *[d_beta[i] for i in range(n)])


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
    def __init__(self, node_id, kind, n_frames, sdf_bounding_sphere, sampler_opt, barf_s, barf_e, no_barf):
        super().__init__()
        self.node_id, self.kind, self.class_id = node_id, kind, CLASS_ID[node_id]
        self.spec = FieldSpec(kind)
        self.sdf_bounding_sphere = sdf_bounding_sphere
        cond = 45 if kind == "hand" else 0
        self.implicit_network = ImplicitNet(3, 6, cond, True, "fourier" if kind == "hand" else "barf", barf_s, barf_e,
                                            no_barf)
        self.rendering_network = RenderingNet("pose", self.spec.rin_dim, [256] * 4, True, pose_dim=cond)
        self.density = LaplaceDensity()
        self.ray_sampler = ErrorBoundSampler(sdf_bounding_sphere, inverse_sphere_bg=True, **sampler_opt)
        self.field = None

    def _field(self, device):
        if self.field is None or self.field.device != device:
            self.field = NodeField(self.spec, device)
        return self.field

    def step_embedding(self):
        self.implicit_network.embedder_obj.step()

    def _eik_field(self, device):
        f = getattr(self, "_eik", None)
        if f is None or f.device != device:
            f = self._eik = NodeField(self.spec, device)
        return f

    def eikonal_grad(self, points):
        """grad_theta of the reference (volsdf_utils.compute_gradient_samples): d sdf / d x at canonical sample
        points [B, n, 3] -> [B, n, 3], differentiable w.r.t. the implicit network's parameters."""
        B, n, _ = points.shape
        xc = torch.zeros(B * n, 4, device=points.device)
        xc[:, :3] = points.reshape(-1, 3)
        barf_w = self.implicit_network.embedder_obj.weights(points.device)
        g = _EikonalFn.apply(self, xc, barf_w, *self._weights())
        return g.view(B, n, 3)

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
        # ---- sampler (no grad; sampler toggles net.eval()/train() in the reference, a no-op for these nets) ----
        if z_override is None:
            with torch.no_grad():
                pk = pack_weights(self.spec, weights[0:9], weights[9:18], weights[18:23], weights[23:28], False)
                dfm = dict(dfm_const)
                dfm["tfs"] = tfs.detach().reshape(-1, nb, 16).contiguous()

                def sdf_query(x, P, out):
                    field.sdf_only(pk, x, P, P // tfs.shape[0], dfm, barf_w, out)

                z_vals = self.ray_sampler.get_z_vals(sdf_query, ray_dirs, cam_loc, self.density.get_beta().item(),
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
    def __init__(self, node_id, betas, n_frames, sdf_bounding_sphere, sampler_opt, mano_model, barf_s, barf_e,
                 no_barf):
        super().__init__(node_id, "hand", n_frames, sdf_bounding_sphere, sampler_opt, barf_s, barf_e, no_barf)
        self.is_rhand = node_id == "right"
        self.server = MANOServer(betas, self.is_rhand, mano_model)
        self.params = GenericParams(n_frames, {"betas": 10, "global_orient": 3, "transl": 3, "pose": 45}, node_id)
        # canonical vertices / skinning table of the KNN deformer (mano/deformer.py:20-32)
        self.register_buffer("cano_verts", self.server.verts_c[0].clone(), persistent=False)

    def sample_eikonal_points(self, batch_size, num=256, local_sigma=0.008, global_ratio=0.20):
        """PointInSpace(global_sigma_xyz=[0.15, 0.06, 0.12]).get_points around random canonical MANO vertices
        (hold_utils.py:22-58,230-240; volsdf_utils.py:28-35): num local + num*ratio global samples."""
        dev = self.cano_verts.device
        idx = torch.randperm(self.cano_verts.shape[0])[:num].to(dev)
        v = self.cano_verts[idx][None].expand(batch_size, -1, -1)
        local = v + torch.randn_like(v) * local_sigma
        gs = torch.tensor([0.15, 0.06, 0.12], device=dev)
        glob = torch.rand(batch_size, int(num * global_ratio), 3, device=dev) * (gs * 2) - gs
        return torch.cat([local, glob], dim=1)

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
    def __init__(self, node_id, n_frames, sdf_bounding_sphere, sampler_opt, entity, barf_s, barf_e, no_barf):
        super().__init__(node_id, "object", n_frames, sdf_bounding_sphere, sampler_opt, barf_s, barf_e, no_barf)
        self.server = ObjectServer(entity)
        self.params = GenericParams(n_frames, {"global_orient": 3, "transl": 3}, node_id)
        self.frame_latent_encoder = nn.Embedding(n_frames, 32)

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
        G.gemm_nt(h[7], W[8], rin[:, 59:59 + FEAT], bias=b[8], N=257, n_split=256, out_raw=sdf)
        dirs = pool.get("dirs", P, 4)
        K.frame_bcast(ray_dirs, P, S, dirs, 0)
        K.embed_fwd(dirs, 3, 4, P, rin, cond=latent, pts_per_frame=ppf)
        R0 = torch.zeros(128, self.Kr, device=dev)
        R0[:, :315] = rw[0]
        R1 = rw[1].contiguous()
        r0 = pool.get("r0", P, 128)
        G.gemm_nt(rin, R0, r0, bias=rb[0].contiguous(), epi=G.EPI_RELU, K=self.Kr)
        rgb = pool.get("rgb", P, 4)
        G.gemm_nt(r0, R1, rgb, bias=rb[1].contiguous(), epi=G.EPI_SIGMOID, N=3)
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
        dR1, dR1b = torch.zeros(3, 128, device=dev), torch.zeros(3, device=dev)
        G.wgrad(dy, sv["r0"], dR1, dR1b, N=3, K=128)
        RT1 = torch.zeros(128, 4, device=dev)
        RT1[:, :3] = sv["R1"].t()
        rr0 = pool.get("rr0", P, 128)
        G.gemm_nt(dy, RT1, rr0, epi=G.EPI_MUL_DRELU, aux1=sv["r0"], K=4)
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
        dW = [torch.zeros_like(w) for w in W]
        dWb = [torch.zeros(w.shape[0], device=dev) for w in W]
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
    """code/src/datasets/utils.py:230-282 (pose-matrix branch); per-pixel elementwise algebra on [B,P]."""
    cam_loc = pose[:, :3, 3]
    fx, fy = intrinsics[:, 0, 0, None], intrinsics[:, 1, 1, None]
    cx, cy, sk = intrinsics[:, 0, 2, None], intrinsics[:, 1, 2, None], intrinsics[:, 0, 1, None]
    x, y = uv[:, :, 0], uv[:, :, 1]
    z = torch.ones_like(x)
    xl = (x - cx + cy * sk / fy - sk * y / fy) / fx * z
    yl = (y - cy) / fy * z
    pc = torch.stack((xl, yl, z, torch.ones_like(z)), -1).permute(0, 2, 1)
    world = torch.bmm(pose, pc).permute(0, 2, 1)[:, :, :3]
    return F.normalize(world - cam_loc[:, None, :], dim=2), cam_loc


DEFAULT_SAMPLER = dict(near=0.0, N_samples=64, N_samples_eval=128, N_samples_extra=32, eps=0.1, beta_iters=10,
                       max_total_iters=5, N_samples_inverse_sphere=32, add_tiny=1e-6)


class HOLDNet(nn.Module):
    """HOLDNet(opt, betas_r, betas_l, num_frames, args) of code/src/hold/hold_net.py:23-51, with the file-backed
    inputs (MANO pickle, data.npy entities) passed explicitly."""

    def __init__(self, scene_bounding_sphere, betas_r, betas_l, num_frames, entities, mano_models, sampler_opt=None,
                 barf_s=1000, barf_e=10000, no_barf=False):
        super().__init__()
        self.sdf_bounding_sphere = float(scene_bounding_sphere)
        self.threshold = 0.05
        so = dict(DEFAULT_SAMPLER if sampler_opt is None else sampler_opt)
        nodes = {}
        if betas_r is not None:
            nodes["right"] = MANONode("right", betas_r, num_frames, self.sdf_bounding_sphere, so, mano_models["right"],
                                      barf_s, barf_e, no_barf)
        if betas_l is not None:
            nodes["left"] = MANONode("left", betas_l, num_frames, self.sdf_bounding_sphere, so, mano_models["left"],
                                     barf_s, barf_e, no_barf)
        nodes["object"] = ObjectNode("object", num_frames, self.sdf_bounding_sphere, so, entities["object"], barf_s,
                                     barf_e, no_barf)
        self.nodes = nn.ModuleDict(nodes)
        self.background = Background(num_frames, self.sdf_bounding_sphere)

    def step_embedding(self):
        for node in self.nodes.values():
            node.step_embedding()
        self.background.step_embedding()

    def forward(self, input, rng=None, z_override=None):
        if not torch.cuda.is_available():
            raise RuntimeError("hold_amd.HOLDNet needs an MI355X: the hot path has no CPU / eager fallback")
        training = self.training
        with torch.enable_grad() if training else torch.no_grad():
            ray_dirs, cam = get_camera_params(input["uv"], input["extrinsics"], input["intrinsics"])
            B, Pn, _ = ray_dirs.shape
            cam_loc = cam.unsqueeze(1).repeat(1, Pn, 1).reshape(-1, 3).contiguous()
            ray_dirs = ray_dirs.reshape(-1, 3).contiguous()
            N = B * Pn
            out = {}
            if training:
                out["epoch"], out["step"] = input["current_epoch"], input["global_step"]
            fac = {}
            for nid, node in self.nodes.items():
                fac[nid] = node.render(input, ray_dirs, cam_loc, Pn, None if rng is None else rng.get(nid),
                                       None if z_override is None else z_override[nid])
            ids = list(fac.keys())
            S = fac[ids[0]]["z_vals"].shape[1]
            args = ([fac[i]["z_vals"] for i in ids] + [fac[i]["sdf"] for i in ids] + [fac[i]["color"] for i in ids] +
                    [fac[i]["normal"] for i in ids] + [self.nodes[i].density.get_beta() for i in ids])
            res = _CompositeFn.apply(S, N, [self.nodes[i].class_id for i in ids], True, *args)
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
            if training:
                self.step_embedding()
        return out
