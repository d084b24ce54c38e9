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


FACES_PER_PIXEL = 100  # RasterizationSettings(faces_per_pixel=100), fitting/utils.py:107


def max_faces_per_pixel(v3d_c, faces, fx, fy, cx, cy, H, W, blur=BLUR):
    """largest number of faces contributing to any pixel (hold_silhouette_max_faces) -- one host read."""
    v = v3d_c.detach().contiguous().float()
    f = faces.to(torch.int32).contiguous()
    B, V, _ = v.shape
    ws = torch.empty(int(_lib.lib().hold_silhouette_workspace_floats(B, f.shape[0])), device=v.device)
    cnt = torch.zeros(1, dtype=torch.int32, device=v.device)
    call("hold_silhouette_max_faces", ptr(v), B, V, ptr(f), f.shape[0], float(fx), float(fy), float(cx), float(cy), int(H), int(W),
         float(blur), ptr(ws), ptr(cnt))
    return int(cnt)


def check_faces_per_pixel(v3d_c, faces, fx, fy, cx, cy, H, W, blur=BLUR):
    """raise if pytorch3d's K = 100 nearest-faces cap would be active (then the uncapped product differs from it)."""
    k = max_faces_per_pixel(v3d_c, faces, fx, fy, cx, cy, H, W, blur)
    if k >= (1 << 30):
        raise NotImplementedError("a face straddles the image plane (vertices behind AND in front of the camera): the reference "
                                  "rasteriser drops such a face per pixel on the interpolated depth (pytorch3d rasterize_meshes), "
                                  "hold_silhouette_fwd drops it as a whole -- the two agree only for meshes in front of the camera")
    if k > FACES_PER_PIXEL:
        raise NotImplementedError(f"{k} faces overlap one pixel: the reference rasteriser keeps only the {FACES_PER_PIXEL} "
                                  "nearest (fitting/utils.py:107); hold_silhouette_fwd multiplies over all of them")
    return k


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


# ------------------------------------------------------------------------------------------ reference call surface
SEGM_IDS = {"bg": 0, "object": 50, "right": 150, "left": 250}  # code/src/utils/const.py:1


def construct_targets(target_masks):
    """fitting/utils.py:161-166"""
    return {k: (target_masks == SEGM_IDS[k]).float() for k in ("object", "right", "left")}


def scaling_masks_K(masks, K, target_dim):
    """fitting/utils.py:189-212: nearest-neighbour resize so the longer side is target_dim, intrinsics scaled alike."""
    B, im_h, im_w = masks.shape
    k = target_dim / max(im_h, im_w)
    masks = torch.nn.functional.interpolate(masks[:, None], size=(int(im_h * k), int(im_w * k)), mode="nearest")[:, 0]
    K4 = torch.eye(4, device=K.device)
    K4[:3, :3] = K[0]
    Ks = torch.diag(torch.tensor([k, k, 1.0, 1.0], device=K.device)) @ K4
    return masks, Ks[None].repeat(B, 1, 1)


class MyParameterDict(torch.nn.ParameterDict):
    """fitting/utils.py:281-294"""

    def search(self, keyword):
        sub = MyParameterDict()
        for key, value in self.items():
            if keyword in key:
                sub[key] = value
        return sub

    def fuzzy_get(self, keyword):
        for k, v in self.items():
            if keyword in k:
                return v
        return None


def _forward_param(server, pd):
    """GenericServer.forward_param / ObjectServer.forward_param (mano/server.py:101-113, obj/server.py:49-56)."""
    go, transl = pd.fuzzy_get("global_orient"), pd.fuzzy_get("transl")
    B = go.shape[0]
    scale = pd.fuzzy_get("scene_scale").view(-1).repeat(B)
    pose = pd.fuzzy_get("__pose")
    if pose is None:
        return server.forward(scale, transl, go)
    return server.forward(scale, transl, torch.cat((go, pose), dim=1), pd.fuzzy_get("betas").repeat(B, 1))


class Model(torch.nn.Module):
    """``Model`` of code/src/fitting/model.py:30-200 with the reference's constructor and methods: any set of nodes
    (right / left / object), every entry of ``param_dict`` a Parameter (keys ``model.nodes.<entity>.params.<name>.weight``
    -> ``<entity>__<name>``), ``obj_scale`` learnable, the loss picked from the node set (:64-78), Adam(1e-2) +
    ReduceLROnPlateau(patience 30) (:146-151), ``fit`` stopping on NaN or lr < 1e-5 (:161-200).  Rendering, the MANO
    LBS and the nearest-neighbour contact term run on the HIP kernels."""

    def __init__(self, servers, scene_scale, obj_scale, param_dict, device, target_masks, w2c, K, fnames, faces,
                 contact_idx=None):
        super().__init__()
        self.w2c, self.servers, self.faces, self.fnames = w2c, servers, faces, fnames
        self.imsize = (target_masks.shape[1], target_masks.shape[2])
        self.node_ids = list(servers.keys())
        self.scene_scale = scene_scale.clone().to(device)
        self.obj_scale = torch.nn.Parameter(torch.as_tensor(np.array(obj_scale), dtype=torch.float32).clone().to(device))
        new = {}
        for key, val in param_dict.items():
            parts = key.split(".")
            new[f"{parts[2]}__{parts[4]}"] = torch.nn.Parameter(val)
        for node_id in servers.keys():
            new[f"{node_id}__scene_scale"] = torch.nn.Parameter(self.scene_scale)
        self.param_dict = MyParameterDict(new)
        self.targets = construct_targets(target_masks)
        self.K = K.clone()
        if contact_idx is None:  # fitting.py:10-13 / loss.py:29-32 read it from ./body_models/contact_zones.pkl
            import pickle
            with open("./body_models/contact_zones.pkl", "rb") as f:
                cz = pickle.load(f)["contact_zones"]
            contact_idx = np.array([i for sub in cz.values() for i in sub])
        self.contact_idx = torch.as_tensor(np.asarray(contact_idx), dtype=torch.long, device=device)
        if "left" in self.node_ids and "right" in self.node_ids:
            self.loss_fn = lambda out, tg: loss_fn_ih(out, tg, self.contact_idx)
        elif "left" in self.node_ids:
            self.loss_fn = lambda out, tg: loss_fn_h(out, tg, "left", self.contact_idx)
        elif "right" in self.node_ids:
            self.loss_fn = lambda out, tg: loss_fn_h(out, tg, "right", self.contact_idx)
        else:
            raise AssertionError(f"Unknown node ids: {self.node_ids}")
        self.pbar = None
        self._k_checked = False

    def freeze_all(self):
        for p in self.parameters():
            p.requires_grad = False

    def defrost_all(self):
        for p in self.parameters():
            p.requires_grad = True

    def print_requires_grad(self):
        print("requires_grad status:")
        for n, p in self.named_parameters():
            print(f"\t{n}: {p.requires_grad}")

    def fwd_params(self):
        from .xdict import xdict
        H, W = self.imsize
        K0 = self.K[0] if self.K.dim() == 3 else self.K
        fx, fy, cx, cy = float(K0[0, 0]), float(K0[1, 1]), float(K0[0, 2]), float(K0[1, 2])
        R, T = self.w2c[:, :3, :3], self.w2c[:, :3, 3:]
        out_dict = xdict()
        self.servers["object"].object_model.obj_scale = self.obj_scale
        for node_id in self.node_ids:
            out = dict(_forward_param(self.servers[node_id], self.param_dict.search(node_id)))
            v3d_c = rigid_tf(out["verts"], R, T)
            out["v3d_c"] = v3d_c
            if node_id in ("right", "left"):
                v_s, f_s = seal_mano_mesh(v3d_c, self.faces[node_id], node_id == "right")
            else:
                v_s, f_s = v3d_c, self.faces[node_id]
            if not self._k_checked:  # once per batch: the K = 100 cap of the reference rasteriser must be inactive
                check_faces_per_pixel(v_s, f_s, fx, fy, cx, cy, H, W)
            out["mask"] = soft_silhouette(v_s, f_s, fx, fy, cx, cy, H, W)
            out_dict.merge(xdict(out).prefix(node_id + "."))
        out_dict["K"] = self.K.clone()
        self._k_checked = True
        return out_dict

    def forward(self):
        out = self.fwd_params()
        return self.loss_fn(out, self.targets), out

    def setup_optimizer(self):
        self.optimizer = torch.optim.Adam(self.parameters(), lr=1e-2)
        self.scheduler = torch.optim.lr_scheduler.ReduceLROnPlateau(self.optimizer, patience=30)

    def fit(self, num_iterations=200, vis_every=50, write_gif=False, out_ps=None):
        """model.py:161-200 (gif writing is visual QA outside the path and is not reproduced).  Returns the loss history."""
        tol_lr = 1e-5
        hist = []
        for _ in range(num_iterations):
            self.optimizer.zero_grad()
            loss_dict, _ = self()
            loss = loss_dict["loss"]
            if torch.isnan(loss) > 0:
                break
            loss.backward()
            self.optimizer.step()
            self.scheduler.step(loss)
            hist.append(float(loss))
            if self.optimizer.param_groups[0]["lr"] < tol_lr:
                break
        return hist


def extract_batch_data(batch_idx, out, masks, device):
    """fitting/utils.py:297-330 with the mask images already decoded (``masks`` [n_frames,H,W] of SEGM ids; the reference
    opens PNGs here): per-batch masks, scene scale, the batch's rows of every parameter, file names, world->camera."""
    idx = np.asarray(batch_idx)
    masks_batch = torch.as_tensor(np.asarray(masks)[idx], dtype=torch.float32).to(device)
    scene_scale = out["scene_scale"].to(device)
    pd = {k: (v[idx].to(device) if ".betas" not in k else v.to(device)) for k, v in out["param_dict"].items()}
    return masks_batch, scene_scale, pd, [out["fnames"][i] for i in batch_idx], out["w2c"].repeat(len(batch_idx), 1, 1).to(device)


def optimize_batch(batch_idx, args, pbar, out, device, obj_scale=None, freeze_scale=False, freeze_shape=False, masks=None,
                   contact_idx=None):
    """code/src/fitting/fitting.py:22-76.  ``out``: {servers, faces, K, w2c, scene_scale, param_dict, fnames}; ``masks``:
    decoded SEGM-id masks of all frames (the reference reads them from disk next to out["fnames"])."""
    masks_batch, scene_scale, param_batch, fnames_batch, w2c_batch = extract_batch_data(batch_idx, out, masks, device)
    masks_batch, K_scaled = scaling_masks_K(masks_batch, out["K"], target_dim=300)
    model = Model(out["servers"], scene_scale, obj_scale, param_batch, device, masks_batch, w2c_batch, K_scaled,
                  fnames_batch, out["faces"], contact_idx=contact_idx)
    model.pbar = pbar
    model.defrost_all()
    model.obj_scale.requires_grad = not freeze_scale
    for k in model.param_dict.keys():
        if "betas" in k and freeze_shape:
            model.param_dict[k].requires_grad = False
        if "pose" in k:
            model.param_dict[k].requires_grad = False
        if "global_orient" in k and "object" not in k:
            model.param_dict[k].requires_grad = False
        if "scene_scale" in k:
            model.param_dict[k].requires_grad = False
    model.setup_optimizer()
    model.history = model.fit(num_iterations=getattr(args, "iters", 300), vis_every=getattr(args, "vis_every", 50),
                              write_gif=False)
    return model
