"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the pose-refinement inner loop's arithmetic
(code/src/fitting/model.py:109-144, code/src/fitting/loss.py:84-165, code/src/fitting/utils.py:101-158).

The silhouette renderer in the reference is pytorch3d 0.7.4 (MeshRasterizer + SoftSilhouetteShader), which is NOT
available here and whose source is not in /root/reference: **parity of this part is unpinned**.  The restatement
follows pytorch3d's documented behaviour:
  * PerspectiveCameras(in_ndc=False, R=diag(-1,-1,1), T=0, K with rows 2/3 swapped): a camera-space point (x,y,z)
    (OpenCV axes) lands at pixel coordinates u = fx x/z + cx, v = fy y/z + cy; NDC = -(u - W/2)/s, -(v - H/2)/s with
    s = min(H,W)/2; pixel (i,j) is sampled at its centre u = j + 0.5, v = i + 0.5.
  * rasterize_meshes(blur_radius, faces_per_pixel=100, bin_size=-1): per pixel, every face whose signed squared NDC
    distance (negative inside, distance to the nearest edge segment) is < blur_radius, in front of the camera.
    (the K = 100 cap only binds when more than 100 faces overlap a pixel; not modelled)
  * sigmoid_alpha_blend: alpha = 1 - prod_f (1 - sigmoid(-d_f / sigma)).

The LOSS functions (loss_fn_h, loss_fn_ih: knn_points K=1, l1, project2d) ARE pinned: scripts/make_golden_fitting.py
imports the reference's own code/src/fitting/loss.py under the shim and records its outputs and gradients in
tests/golden/fitting_losses.npz; tests/test_oracle_golden.py checks this restatement against them (identical).
"""
from __future__ import annotations

import math

import numpy as np
import torch

SEAL_FACES_R = [[120, 108, 778], [108, 79, 778], [79, 78, 778], [78, 121, 778], [121, 214, 778], [214, 215, 778],
                [215, 279, 778], [279, 239, 778], [239, 234, 778], [234, 92, 778], [92, 38, 778], [38, 122, 778],
                [122, 118, 778], [118, 117, 778], [117, 119, 778], [119, 120, 778]]
CIRCLE_V_ID = [108, 79, 78, 121, 214, 215, 279, 239, 234, 92, 38, 122, 118, 117, 119, 120]
SIGMA = 1e-6
BLUR = math.log(1.0 / 1e-4 - 1.0) * SIGMA


def seal_mano_mesh(v3d, faces, is_rhand):
    """common/body_models.py:62-73."""
    seal = torch.tensor(SEAL_FACES_R, dtype=torch.long)
    if not is_rhand:
        seal = seal[:, [1, 0, 2]]
    centers = v3d[:, CIRCLE_V_ID].mean(dim=1)[:, None, :]
    return torch.cat((v3d, centers), dim=1), torch.cat((faces, seal), dim=0)


def rigid_tf(points, R, T):
    """common/transforms.py:137-148."""
    return (torch.bmm(R, points.permute(0, 2, 1)) + T).permute(0, 2, 1)


def to_ndc(v3d_c, fx, fy, cx, cy, H, W):
    s = min(H, W) / 2.0
    u = fx * v3d_c[..., 0] / v3d_c[..., 2] + cx
    v = fy * v3d_c[..., 1] / v3d_c[..., 2] + cy
    return torch.stack([-(u - W / 2.0) / s, -(v - H / 2.0) / s], -1)


def _seg_d2(p, a, b):
    ab = b - a
    t = ((p - a) * ab).sum(-1) / (ab * ab).sum(-1).clamp_min(1e-20)
    t = t.clamp(0, 1)
    q = a + t[..., None] * ab
    return ((p - q) ** 2).sum(-1)


def soft_silhouette(v3d_c, faces, fx, fy, cx, cy, H, W, sigma=SIGMA, blur=BLUR, chunk=2048, return_count=False, cull="pixel"):
    """v3d_c [B,V,3] camera-space vertices, faces [F,3] -> alpha [B,H,W] (and, with return_count, the number of faces
    contributing to each pixel [B,H,W]: the rasteriser of the reference keeps at most faces_per_pixel = 100 of them,
    fitting/utils.py:107 -- the product over all faces below equals it while that count stays <= 100).
    cull: "pixel" (default) = pytorch3d's rule as published (rasterize_meshes.cu, CheckPixelInsideFace / the naive kernel): a
    face whose largest vertex depth is < 0 is skipped, and a (pixel, face) pair is skipped when the depth interpolated with
    the perspective-corrected, clipped barycentric coordinates of the pixel is < 0 (perspective_correct and
    clip_barycentric_coords both default to True for a perspective camera with blur_radius > 0); "face" = the HIP kernel's
    rule (skip a face unless ALL its vertices are in front of the camera).  The two agree for meshes in front of the camera
    (tests/test_host_cpu.py); hold_amd.fitting.check_faces_per_pixel refuses the configurations where they do not."""
    B = v3d_c.shape[0]
    ndc = to_ndc(v3d_c, fx, fy, cx, cy, H, W)  # [B,V,2]
    z = v3d_c[..., 2]
    s = min(H, W) / 2.0
    jj, ii = torch.meshgrid(torch.arange(W, dtype=v3d_c.dtype), torch.arange(H, dtype=v3d_c.dtype), indexing="xy")
    px = -((jj + 0.5) - W / 2.0) / s
    py = -((ii + 0.5) - H / 2.0) / s
    pix = torch.stack([px, py], -1).reshape(-1, 2)  # [HW,2]
    out, counts = [], []
    for b in range(B):
        cnt = torch.zeros(pix.shape[0], dtype=torch.long)
        tri = ndc[b][faces]  # [F,3,2]
        zf = z[b][faces]  # [F,3]
        logacc = torch.zeros(pix.shape[0], dtype=v3d_c.dtype)
        for c0 in range(0, pix.shape[0], chunk):
            p = pix[c0:c0 + chunk][:, None, :]  # [P,1,2]
            a, bb, cc = tri[None, :, 0], tri[None, :, 1], tri[None, :, 2]
            e0 = (bb[..., 0] - a[..., 0]) * (p[..., 1] - a[..., 1]) - (bb[..., 1] - a[..., 1]) * (p[..., 0] - a[..., 0])
            e1 = (cc[..., 0] - bb[..., 0]) * (p[..., 1] - bb[..., 1]) - (cc[..., 1] - bb[..., 1]) * (p[..., 0] - bb[..., 0])
            e2 = (a[..., 0] - cc[..., 0]) * (p[..., 1] - cc[..., 1]) - (a[..., 1] - cc[..., 1]) * (p[..., 0] - cc[..., 0])
            inside = ((e0 >= 0) & (e1 >= 0) & (e2 >= 0)) | ((e0 <= 0) & (e1 <= 0) & (e2 <= 0))
            area = (bb[..., 0] - a[..., 0]) * (cc[..., 1] - a[..., 1]) - (bb[..., 1] - a[..., 1]) * (cc[..., 0] - a[..., 0])
            d2 = torch.minimum(torch.minimum(_seg_d2(p, a, bb), _seg_d2(p, bb, cc)), _seg_d2(p, cc, a))
            d = torch.where(inside, -d2, d2)
            valid = (inside | (d2 < blur)) & (area.abs() > 1e-8)
            if cull == "face":
                valid = valid & (zf.min(-1).values[None] > 0)
            else:
                z0, z1, z2 = zf[None, :, 0], zf[None, :, 1], zf[None, :, 2]
                sa = torch.where(area.abs() > 1e-8, area, torch.ones_like(area))
                w0, w1, w2 = e1 / sa, e2 / sa, e0 / sa  # barycentric weights of a, b, c (unclipped)
                t0, t1, t2 = w0 * z1 * z2, z0 * w1 * z2, z0 * z1 * w2  # BarycentricPerspectiveCorrectionForward
                den = torch.clamp(t0 + t1 + t2, min=1e-8)
                q0, q1, q2 = (torch.clamp(t / den, min=0.0) for t in (t0, t1, t2))  # BarycentricClipForward (lower bound only)
                qs = torch.clamp(q0 + q1 + q2, min=1e-5)
                pz = (q0 * z0 + q1 * z1 + q2 * z2) / qs
                valid = valid & (zf.max(-1).values[None] >= 0) & (pz >= 0)
            # 1 - prob = sigmoid(d / sigma)
            l1mp = torch.nn.functional.logsigmoid(d / sigma)
            logacc[c0:c0 + chunk] = torch.where(valid, l1mp, torch.zeros_like(l1mp)).sum(-1)
            cnt[c0:c0 + chunk] = valid.sum(-1)
        out.append((1.0 - torch.exp(logacc)).reshape(H, W))
        counts.append(cnt.reshape(H, W))
    if return_count:
        return torch.stack(out), torch.stack(counts)
    return torch.stack(out)


def knn1_mean(q, t):
    """knn_points(q, t, K=1)[0] (squared distances [B,Nq,1])."""
    d = ((q[:, :, None, :] - t[:, None, :, :]) ** 2).sum(-1)
    return d.min(-1).values


def loss_fn_h(out, targets, flag, contact_idx):
    """code/src/fitting/loss.py:84-110."""
    tips = out[f"{flag}.v3d_c"][:, contact_idx]
    fine = knn1_mean(tips, out["object.v3d_c"]).mean()
    vp = 1 - targets[flag]
    lo = ((out["object.mask"] - targets["object"]).abs() * vp).sum() / vp.sum()
    vp = 1 - targets["object"]
    lh = ((out[f"{flag}.mask"] - targets[flag]).abs() * vp).sum() / vp.sum()
    return {"mask_o": lo * 1000, "mask_h": lh * 1000, "fine_ho": fine * 100.0, "loss": lo * 1000 + lh * 1000 + fine * 100.0}


def project2d(K, pts_cam):
    """project2d_batch, common/transforms.py:339-352."""
    Kb = K if K.dim() == 3 else K[None].expand(pts_cam.shape[0], -1, -1)
    h = torch.bmm(Kb, pts_cam.permute(0, 2, 1)).permute(0, 2, 1)
    return h[..., :2] / h[..., 2:3]


def loss_fn_ih(out, targets, contact_idx):
    """code/src/fitting/loss.py:120-165 (statement by statement, incl. the in-place thresholding)."""
    valid_pix = (1 - targets["right"]) * (1 - targets["left"])
    err = (out["object.mask"] - targets["object"]).abs() * valid_pix
    loss_mask_o = err.sum() / valid_pix.sum()
    dist_thres = 2.0 ** 2
    c_ro = knn1_mean(out["right.v3d_c"][:, contact_idx], out["object.v3d_c"]).mean(dim=1)
    c_lo = knn1_mean(out["left.v3d_c"][:, contact_idx], out["object.v3d_c"]).mean(dim=1)
    c_ro = c_ro.clone()
    c_lo = c_lo.clone()
    c_ro[c_ro < dist_thres] = 0
    c_lo[c_lo < dist_thres] = 0
    j2d_r = project2d(out["K"], out["right.v3d_c"])
    j2d_l = project2d(out["K"], out["left.v3d_c"])
    if "j2d_r_target" not in targets:
        targets["j2d_r_target"] = j2d_r.detach().clone()
        targets["j2d_l_target"] = j2d_l.detach().clone()
    d = {"mask_o": loss_mask_o * 1000,
         "v2d_r": torch.nn.functional.mse_loss(j2d_r, targets["j2d_r_target"]),
         "v2d_l": torch.nn.functional.mse_loss(j2d_l, targets["j2d_l_target"]),
         "contact_ro": c_ro.mean() * 0.05, "contact_lo": c_lo.mean() * 0.05}
    d["loss"] = sum(d.values())
    return d
