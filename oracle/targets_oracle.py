"""TEST INFRASTRUCTURE ONLY -- CPU restatement (torch / numpy) of the training-only side of the hot path:

* ``prepare_loss_targets_hand / _object``      code/src/hold/hold_utils.py:149-240 (via volsdf_utils.py:19-48,172-217)
* ``spawn_cano_mano`` = seal + one Loop step    code/src/model/renderables/mano_node.py:126-135, hold_utils.py:137-146
* ``Loss.forward`` and its terms                code/src/hold/loss.py:17-93, code/src/hold/loss_terms.py:14-111

Random draws are inputs here (the product records the points it sampled), so comparisons are deterministic.
kaolin's point->mesh distance / inside test are restated by exact geometry in oracle/geometry_oracle.py and trimesh's
Loop subdivision by Loop's rules below (neither package is available: parity with them is unpinned, see those headers).
``loss_forward`` is pinned against the reference's own ``Loss`` in tests/test_dropin_cpu.py (build container only).
"""
from __future__ import annotations

import math

import numpy as np
import torch

from . import geometry_oracle as go
from . import hold_oracle as ho


# ------------------------------------------------------------------------------------------ Loop subdivision
def subdivide_loop(verts, faces):
    """one Loop subdivision step with explicit per-edge / per-vertex loops (numpy, fp64).  Returns (verts, faces) with
    the vertex order [relaxed originals, one vertex per edge in order of first appearance over faces x (01, 12, 20)]."""
    v = np.asarray(verts, dtype=np.float64)
    f = np.asarray(faces, dtype=np.int64)
    edge_id, edge_opp, edge_ends = {}, [], []
    for tri in f:
        for a, b, c in ((tri[0], tri[1], tri[2]), (tri[1], tri[2], tri[0]), (tri[2], tri[0], tri[1])):
            key = (min(a, b), max(a, b))
            if key not in edge_id:
                edge_id[key] = len(edge_ends)
                edge_ends.append(key)
                edge_opp.append([])
            edge_opp[edge_id[key]].append(c)
    nbr = [set() for _ in range(len(v))]
    for a, b in edge_ends:
        nbr[a].add(b)
        nbr[b].add(a)
    boundary = [len(o) == 1 for o in edge_opp]
    odd = np.zeros((len(edge_ends), 3))
    for e, (a, b) in enumerate(edge_ends):
        if boundary[e]:
            odd[e] = 0.5 * (v[a] + v[b])
        else:
            odd[e] = 0.375 * (v[a] + v[b]) + 0.125 * (v[edge_opp[e][0]] + v[edge_opp[e][1]])
    bnd_nbr = [[] for _ in range(len(v))]
    for e, (a, b) in enumerate(edge_ends):
        if boundary[e]:
            bnd_nbr[a].append(b)
            bnd_nbr[b].append(a)
    even = np.zeros_like(v)
    for i in range(len(v)):
        if bnd_nbr[i]:
            even[i] = 0.75 * v[i] + 0.125 * sum(v[j] for j in bnd_nbr[i])
            continue
        k = len(nbr[i])
        if k == 0:
            even[i] = v[i]
            continue
        beta = (1.0 / k) * (5.0 / 8.0 - (3.0 / 8.0 + 0.25 * math.cos(2 * math.pi / k)) ** 2)
        even[i] = (1 - k * beta) * v[i] + beta * sum(v[j] for j in nbr[i])
    nv = len(v)
    new_faces = []
    for tri in f:
        e01 = nv + edge_id[(min(tri[0], tri[1]), max(tri[0], tri[1]))]
        e12 = nv + edge_id[(min(tri[1], tri[2]), max(tri[1], tri[2]))]
        e20 = nv + edge_id[(min(tri[2], tri[0]), max(tri[2], tri[0]))]
        new_faces += [[tri[0], e01, e20], [e01, tri[1], e12], [e20, e12, tri[2]], [e01, e12, e20]]
    return np.concatenate([even, odd], 0), np.array(new_faces, dtype=np.int64)


def mesh_as_triangle_set(verts, faces, decimals=6):
    """order-independent description of a mesh: sorted canonical triangles over rounded coordinates."""
    v = np.round(np.asarray(verts, dtype=np.float64), decimals) + 0.0
    out = []
    for t in np.asarray(faces):
        keys = [tuple(v[i]) for i in t]
        k = keys.index(min(keys))
        out.append(tuple(keys[(k + i) % 3] for i in range(3)))
    return sorted(out)


# ------------------------------------------------------------------------------------------ loss targets
def loss_targets_hand(sd, node, mesh_v_div, mesh_f_div, cano_pts, mano_cano_samples, eik_samples, n_pix_total,
                      embed_w=None):
    """prepare_loss_targets_hand (hold_utils.py:186-240) for given sample points.  cano_pts [B, n_pix*S, 3]."""
    B = cano_pts.shape[0]
    mv = mesh_v_div[None].expand(B, -1, -1).double()
    out = {}
    out[f"{node}.pts2mano_sdf_cano"] = go.compute_mano_cano_sdf(mv, mesh_f_div, mano_cano_samples.double()).to(cano_pts.dtype)
    x = mano_cano_samples.reshape(-1, 3)
    cond = torch.zeros(x.shape[0], 45, dtype=x.dtype)
    pred = ho.implicit_net(sd, f"nodes.{node}.implicit_network", x, cond, 6, embed_w, zero_cond=True)[:, 0]
    out[f"{node}.pred_sdf"] = pred.view(B, -1)
    off, _ = go.check_off_in_surface_points_cano_mesh(mv, mesh_f_div, cano_pts.double(), n_pix_total, threshold=0.01)
    out[f"{node}.index_off_surface"] = off
    out[f"{node}.grad_theta"] = grad_theta(sd, node, eik_samples, embed_w)
    return out


def loss_targets_object(sd, node, mesh_vo, mesh_fo, cano_pts, eik_samples, n_pix_total, embed_w=None):
    """prepare_loss_targets_object (hold_utils.py:149-183)."""
    B = cano_pts.shape[0]
    mv = mesh_vo[None].expand(B, -1, -1).double()
    off, _ = go.check_off_in_surface_points_cano_mesh(mv, mesh_fo, cano_pts.double(), n_pix_total, threshold=0.05)
    return {f"{node}.index_off_surface": off, f"{node}.grad_theta": grad_theta(sd, node, eik_samples, embed_w)}


def grad_theta(sd, node, samples, embed_w=None):
    B, n, _ = samples.shape
    x = samples.reshape(-1, 3).detach().clone().requires_grad_(True)
    is_obj = node == "object"
    cond = None if is_obj else torch.zeros(x.shape[0], 45, dtype=x.dtype)
    sdf = ho.implicit_net(sd, f"nodes.{node}.implicit_network", x, cond, 6, embed_w, zero_cond=not is_obj)[:, :1]
    g = torch.autograd.grad(sdf, x, torch.ones_like(sdf), create_graph=True, retain_graph=True)[0]
    return g.view(B, n, 3)


# ------------------------------------------------------------------------------------------ Loss
def loss_forward(batch, out, milestone=30000):
    """Loss.forward (code/src/hold/loss.py:17-93); image_scores == 1 (:24), valid_pix == 1 (:36)."""
    rgb = out["rgb"]
    rgb_gt = batch["gt.rgb"].reshape(-1, 3)
    mask_gt = batch["gt.mask"].reshape(-1)
    n = float(mask_gt.shape[0])
    nan_filter = ~torch.any(rgb.isnan(), dim=1)
    rgb_loss = (rgb[nan_filter] - rgb_gt[nan_filter]).abs().sum() / (nan_filter.float().sum() + 1e-6)  # loss_terms.py:14-20
    cls = torch.zeros(mask_gt.shape, dtype=torch.long)  # get_sem_loss, loss_terms.py:67-98
    cls[(mask_gt >= 25) & (mask_gt < 100)] = 1
    cls[(mask_gt >= 100) & (mask_gt < 200)] = 2
    cls[mask_gt >= 200] = 3
    onehot = torch.nn.functional.one_hot(cls, 4).to(rgb.dtype)
    sem_loss = ((out["semantics"] - onehot) ** 2).sum() / n
    sparse = 0.0
    for k in list(out.keys()):
        if "index_off_surface" in k:
            nid = k.split(".")[0]
            acc = out[f"{nid}.mask_prob"]
            sparse = sparse + acc[out[k]].abs().mean()  # get_opacity_sparse_loss :44-56
    eik = 0.0
    for k in list(out.keys()):
        if "grad_theta" in k:
            eik = eik + ((out[k].norm(2, dim=-1) - 1) ** 2).mean()  # get_eikonal_loss :24-26
    mano = 0.0
    for k in list(out.keys()):
        if "pts2mano_sdf_cano" in k:
            nid = k.split(".")[0]
            gt = torch.clamp(out[k].detach(), -0.01, 0.01)
            pr = torch.clamp(out[f"{nid}.pred_sdf"], -0.01, 0.01)
            mano = mano + (pr - gt).abs().mean()  # get_mano_cano_loss :101-111
    progress = min(milestone, int(out["step"]))
    w_sem = torch.linspace(1.1, 0.1, milestone + 1)[progress]
    w_sparse = torch.linspace(0.0, 1.0, milestone + 1)[progress]
    ld = {"loss/rgb": rgb_loss * 1.0, "loss/sem": sem_loss * w_sem}
    eik = eik * 0.00001
    if eik > 0.0008:
        ld["loss/eikonal"] = eik
    ld["loss/mano_cano"] = mano * 5.0
    ld["loss/opacity_sparse"] = sparse * w_sparse
    ld["loss"] = sum(ld[k] for k in list(ld.keys()))
    return ld
