"""Pose-refinement inner loop on the HIP kernels -- host mirror of code/src/fitting/model.py (Model.fwd_params,
Model.fit) and code/src/fitting/loss.py (loss_fn_h / loss_fn_ih): per iteration MANO / object forward kinematics
(hold_mano_lbs_fwd/bwd), world->camera rigid transform, sealed-mesh soft silhouettes (hold_silhouette_fwd/bwd),
mask L1 + K=1 contact terms (hold_knn1_fwd/bwd), Adam(lr 1e-2) + ReduceLROnPlateau(patience 30), stop at lr < 1e-5."""
from __future__ import annotations

import math

import numpy as np
import torch

from . import _lib
from ._lib import call, ptr

SIGMA = 1e-6
BLUR = math.log(1.0 / 1e-4 - 1.0) * SIGMA  # fitting/utils.py:101-108
SEAL_FACES_R = [[120, 108, 778], [108, 79, 778], [79, 78, 778], [78, 121, 778], [121, 214, 778], [214, 215, 778],
                [215, 279, 778], [279, 239, 778], [239, 234, 778], [234, 92, 778], [92, 38, 778], [38, 122, 778],
                [122, 118, 778], [118, 117, 778], [117, 119, 778], [119, 120, 778]]
CIRCLE_V_ID = [108, 79, 78, 121, 214, 215, 279, 239, 234, 92, 38, 122, 118, 117, 119, 120]


def seal_mano_mesh(v3d, faces, is_rhand):
    """common/body_models.py:62-73 (wrist cap: one extra vertex at the ring centre, 16 extra faces)."""
    seal = torch.tensor(SEAL_FACES_R, dtype=faces.dtype, device=faces.device)
    if not is_rhand:
        seal = seal[:, [1, 0, 2]]
    centers = v3d[:, CIRCLE_V_ID].mean(dim=1)[:, None, :]
    return torch.cat((v3d, centers), dim=1), torch.cat((faces, seal), dim=0)


class _SilhouetteFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, v3d_c, faces_i32, fx, fy, cx, cy, H, W, sigma, blur):
        B, V, _ = v3d_c.shape
        F = faces_i32.shape[0]
        v = v3d_c.detach().contiguous().float()
        ws = torch.empty(int(_lib.lib().hold_silhouette_workspace_floats(B, F)), device=v.device)
        mask = torch.empty(B, H, W, device=v.device)
        call("hold_silhouette_fwd", ptr(v), B, V, ptr(faces_i32), F, fx, fy, cx, cy, H, W, sigma, blur, ptr(ws), ptr(mask))
        ctx.save_for_backward(v, faces_i32)
        ctx.cfg = (fx, fy, cx, cy, H, W, sigma, blur)
        return mask

    @staticmethod
    def backward(ctx, d_mask):
        v, faces_i32 = ctx.saved_tensors
        fx, fy, cx, cy, H, W, sigma, blur = ctx.cfg
        B, V, _ = v.shape
        F = faces_i32.shape[0]
        ws = torch.empty(int(_lib.lib().hold_silhouette_workspace_floats(B, F)), device=v.device)
        dndc = torch.empty(B, V, 2, device=v.device)
        dv = torch.empty(B, V, 3, device=v.device)
        call("hold_silhouette_bwd", ptr(v), B, V, ptr(faces_i32), F, fx, fy, cx, cy, H, W, sigma, blur, ptr(ws),
             ptr(d_mask.contiguous()), ptr(dndc), ptr(dv))
        return dv, None, None, None, None, None, None, None, None, None


def soft_silhouette(v3d_c, faces, fx, fy, cx, cy, H, W, sigma=SIGMA, blur=BLUR):
    """alpha channel of MeshRenderer(SoftSilhouetteShader) for camera-space vertices [B,V,3]."""
    return _SilhouetteFn.apply(v3d_c, faces.to(torch.int32).contiguous(), float(fx), float(fy), float(cx), float(cy),
                               int(H), int(W), float(sigma), float(blur))


class _Knn1Fn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, q, t):
        B, Nq, _ = q.shape
        Nt = t.shape[1]
        qq, tt = q.detach().contiguous().float(), t.detach().contiguous().float()
        d2 = torch.empty(B, Nq, device=q.device)
        idx = torch.empty(B, Nq, dtype=torch.int32, device=q.device)
        call("hold_knn1_fwd", ptr(qq), B, Nq, ptr(tt), Nt, ptr(d2), ptr(idx))
        ctx.save_for_backward(qq, tt, idx)
        return d2

    @staticmethod
    def backward(ctx, g):
        qq, tt, idx = ctx.saved_tensors
        B, Nq, _ = qq.shape
        dq = torch.empty_like(qq)
        dt = torch.zeros_like(tt)
        call("hold_knn1_bwd", ptr(qq), B, Nq, ptr(tt), tt.shape[1], ptr(idx), ptr(g.contiguous()), ptr(dq), ptr(dt))
        return dq, dt


def knn1_sqdist(q, t):
    """pytorch3d.ops.knn_points(q, t, K=1)[0][..., 0]"""
    return _Knn1Fn.apply(q, t)


def rigid_tf(points, R, T):
    """common/transforms.py:137-148"""
    return (torch.bmm(R, points.permute(0, 2, 1)) + T).permute(0, 2, 1)


def loss_fn_h(out, targets, flag, contact_idx):
    """code/src/fitting/loss.py:84-110"""
    tips = out[f"{flag}.v3d_c"][:, contact_idx]
    fine = knn1_sqdist(tips, out["object.v3d_c"]).mean()
    vp = 1 - targets[flag]
    lo = ((out["object.mask"] - targets["object"]).abs() * vp).sum() / vp.sum()
    vp = 1 - targets["object"]
    lh = ((out[f"{flag}.mask"] - targets[flag]).abs() * vp).sum() / vp.sum()
    d = {"mask_o": lo * 1000, "mask_h": lh * 1000, "fine_ho": fine * 100.0}
    d["loss"] = sum(d.values())
    return d


def project2d(K, pts_cam):
    """project2d_batch (common/transforms.py:339-352): pixel coordinates of camera-space points; K [3,3] or [B,3,3]."""
    Kb = K if K.dim() == 3 else K[None].expand(pts_cam.shape[0], -1, -1)
    h = torch.bmm(Kb, pts_cam.permute(0, 2, 1)).permute(0, 2, 1)
    return h[..., :2] / h[..., 2:3]


def loss_fn_ih(out, targets, contact_idx, dist_thres=2.0 ** 2):
    """two-hand loss, code/src/fitting/loss.py:120-165: object mask L1 on pixels covered by neither hand (x1000), per-frame
    mean fingertip-zone -> object squared distance kept only above dist_thres (x0.05, each hand), and an MSE that pins the
    projected hand vertices to their first-iteration positions (x1; the targets are cached in `targets` on first use)."""
    vp = (1 - targets["right"]) * (1 - targets["left"])
    lo = ((out["object.mask"] - targets["object"]).abs() * vp).sum() / vp.sum()
    vo = out["object.v3d_c"]
    contact = {}
    for fl in ("right", "left"):
        c = knn1_sqdist(out[f"{fl}.v3d_c"][:, contact_idx], vo).mean(dim=1)
        contact[fl] = torch.where(c < dist_thres, torch.zeros_like(c), c).mean()
    j2d = {fl: project2d(out["K"], out[f"{fl}.v3d_c"]) for fl in ("right", "left")}
    if "j2d_r_target" not in targets:
        targets["j2d_r_target"] = j2d["right"].detach().clone()
        targets["j2d_l_target"] = j2d["left"].detach().clone()
    d = {"mask_o": lo * 1000,
         "v2d_r": torch.nn.functional.mse_loss(j2d["right"], targets["j2d_r_target"]),
         "v2d_l": torch.nn.functional.mse_loss(j2d["left"], targets["j2d_l_target"]),
         "contact_ro": contact["right"] * 0.05, "contact_lo": contact["left"] * 0.05}
    d["loss"] = sum(d.values())
    return d


class FittingModel(torch.nn.Module):
    """Model of code/src/fitting/model.py:30-200 for one hand + object (the HO3D / in-the-wild case):
    free parameters = hand translation, object rotation + translation (fitting.py:57-67); everything else frozen."""

    def __init__(self, hand_server, obj_server, hand_faces, obj_faces, params, w2c, K, imsize, targets, contact_idx,
                 flag="right"):
        super().__init__()
        self.flag = flag
        self.hand_server, self.obj_server = hand_server, obj_server
        self.hand_faces, self.obj_faces = hand_faces, obj_faces
        self.w2c, self.K, self.imsize = w2c, K, imsize
        self.targets, self.contact_idx = targets, contact_idx
        self.frozen = {k: v for k, v in params.items() if k not in (f"{flag}.transl", "object.global_orient", "object.transl")}
        self.h_transl = torch.nn.Parameter(params[f"{flag}.transl"].clone())
        self.o_rot = torch.nn.Parameter(params["object.global_orient"].clone())
        self.o_transl = torch.nn.Parameter(params["object.transl"].clone())

    def fwd_params(self):
        f, fl = self.frozen, self.flag
        B = self.h_transl.shape[0]
        H, W = self.imsize
        fx, fy, cx, cy = (float(self.K[0, 0]), float(self.K[1, 1]), float(self.K[0, 2]), float(self.K[1, 2]))
        scale = f["scene_scale"].view(-1).expand(B).contiguous()
        full_pose = torch.cat((f[f"{fl}.global_orient"], f[f"{fl}.pose"]), 1)
        ho = self.hand_server(scale, self.h_transl, full_pose, f[f"{fl}.betas"].expand(B, -1).contiguous())
        oo = self.obj_server(scale, self.o_transl, self.o_rot)
        out = {}
        R, T = self.w2c[:, :3, :3], self.w2c[:, :3, 3:]
        vh = rigid_tf(ho["verts"], R, T)
        vo = rigid_tf(oo["verts"], R, T)
        out[f"{fl}.v3d_c"], out["object.v3d_c"] = vh, vo
        vh_s, fh_s = seal_mano_mesh(vh, self.hand_faces, fl == "right")
        out[f"{fl}.mask"] = soft_silhouette(vh_s, fh_s, fx, fy, cx, cy, H, W)
        out["object.mask"] = soft_silhouette(vo, self.obj_faces, fx, fy, cx, cy, H, W)
        return out

    def forward(self):
        return loss_fn_h(self.fwd_params(), self.targets, self.flag, self.contact_idx)

    def fit(self, num_iterations=300, tol_lr=1e-5):
        """Model.fit (model.py:161-200)"""
        opt = torch.optim.Adam(self.parameters(), lr=1e-2)
        sched = torch.optim.lr_scheduler.ReduceLROnPlateau(opt, patience=30)
        hist = []
        for _ in range(num_iterations):
            opt.zero_grad()
            ld = self()
            loss = ld["loss"]
            if torch.isnan(loss):
                break
            loss.backward()
            opt.step()
            sched.step(loss)
            hist.append(float(loss))
            if opt.param_groups[0]["lr"] < tol_lr:
                break
        return hist


class FittingModelIH(torch.nn.Module):
    """two-hand + object variant of Model (model.py:64-78 picks loss_fn_ih when both hands are present): free parameters =
    both hand translations, object rotation + translation.  The reference also rasterises the two hand masks every
    iteration (model.py:121-140) although loss_fn_ih never reads them; they are skipped here (no effect on the loss or
    its gradients)."""

    def __init__(self, hand_servers, obj_server, obj_faces, params, w2c, K, imsize, targets, contact_idx):
        super().__init__()
        self.hand_servers, self.obj_server, self.obj_faces = hand_servers, obj_server, obj_faces
        self.w2c, self.K, self.imsize = w2c, K, imsize
        self.targets, self.contact_idx = targets, contact_idx
        free = ("right.transl", "left.transl", "object.global_orient", "object.transl")
        self.frozen = {k: v for k, v in params.items() if k not in free}
        self.r_transl = torch.nn.Parameter(params["right.transl"].clone())
        self.l_transl = torch.nn.Parameter(params["left.transl"].clone())
        self.o_rot = torch.nn.Parameter(params["object.global_orient"].clone())
        self.o_transl = torch.nn.Parameter(params["object.transl"].clone())

    def fwd_params(self):
        f = self.frozen
        B = self.r_transl.shape[0]
        H, W = self.imsize
        fx, fy, cx, cy = (float(self.K[0, 0]), float(self.K[1, 1]), float(self.K[0, 2]), float(self.K[1, 2]))
        scale = f["scene_scale"].view(-1).expand(B).contiguous()
        R, T = self.w2c[:, :3, :3], self.w2c[:, :3, 3:]
        out = {"K": self.K}
        for fl, tr in (("right", self.r_transl), ("left", self.l_transl)):
            full_pose = torch.cat((f[f"{fl}.global_orient"], f[f"{fl}.pose"]), 1)
            ho = self.hand_servers[fl](scale, tr, full_pose, f[f"{fl}.betas"].expand(B, -1).contiguous())
            out[f"{fl}.v3d_c"] = rigid_tf(ho["verts"], R, T)
        oo = self.obj_server(scale, self.o_transl, self.o_rot)
        vo = rigid_tf(oo["verts"], R, T)
        out["object.v3d_c"] = vo
        out["object.mask"] = soft_silhouette(vo, self.obj_faces, fx, fy, cx, cy, H, W)
        return out

    def forward(self):
        return loss_fn_ih(self.fwd_params(), self.targets, self.contact_idx)

    fit = FittingModel.fit
