"""Training loss-target geometry on the HIP point->mesh kernel: the kaolin-backed helpers of
code/src/engine/volsdf_utils.py:172-217 and the samplers of code/src/hold/hold_utils.py:22-55,274-303, kaolin-free."""
from __future__ import annotations

import torch

from ._lib import call, ptr


def mesh_sdf(points, verts, faces, cull_dist=0.0):
    """signed distance (negative inside) of points [B,P,3] to the closed mesh verts [B,V,3] | [V,3], faces [F,3].
    cull_dist > 0: points farther than that from the mesh's bounding box get their distance to the box instead."""
    assert points.dim() == 3 and points.shape[-1] == 3
    B, P, _ = points.shape
    pts = points.detach().contiguous().float()
    shared = verts.dim() == 2
    v = verts.detach().contiguous().float()
    V = v.shape[-2]
    assert shared or v.shape[0] == B
    f = faces.to(torch.int32).contiguous()
    out = torch.empty(B, P, device=points.device)
    aabb = None
    if cull_dist > 0:
        vv = v[None].expand(B, -1, -1) if shared else v
        aabb = torch.cat([vv.min(dim=1).values, vv.max(dim=1).values], dim=1).contiguous()
    call("hold_mesh_sdf", ptr(pts), B, P, ptr(v), int(shared), V, ptr(f), f.shape[0], float(cull_dist), ptr(aabb), ptr(out))
    return out


def compute_mano_cano_sdf(mesh_v_cano, mesh_f_cano, x_cano):
    """volsdf_utils.py:172-186 (the kaolin face-vertex tensor argument is not needed)."""
    return mesh_sdf(x_cano, mesh_v_cano, mesh_f_cano)


def check_off_in_surface_points_cano_mesh(mesh_v_cano, mesh_f_cano, x_cano, num_pixels_total, threshold=0.05):
    """volsdf_utils.py:189-217: per ray, min signed distance over its samples -> (index_off_surface, index_in_surface).
    Points farther than 2 x threshold from the mesh's bounding box are culled (they can change neither test)."""
    sd = mesh_sdf(x_cano, mesh_v_cano, mesh_f_cano, cull_dist=2.0 * threshold).reshape(num_pixels_total, -1)
    minimum = sd.min(dim=1).values
    return minimum > threshold, minimum <= 0.0


def sample_on_barycentric_mesh(verts, faces, num_samples):
    """hold_utils.py:274-303 (same draws in the same order: randint faces, rand u, rand v)."""
    B = verts.shape[0]
    fi = torch.randint(0, faces.shape[0], (B, num_samples), device=verts.device)
    sf = faces[fi]
    g = lambda k: torch.gather(verts, 1, sf[..., k].unsqueeze(-1).expand(-1, -1, 3))
    v0, v1, v2 = g(0), g(1), g(2)
    u = torch.rand((B, num_samples, 1), device=verts.device)
    v = torch.rand((B, num_samples, 1), device=verts.device)
    mask = u + v > 1
    u, v = torch.where(mask, 1 - u, u), torch.where(mask, 1 - v, v)
    return u * v0 + v * v1 + (1 - u - v) * v2


class PointInSpace:
    """hold_utils.py:22-55: one Gaussian-perturbed copy of every centre + global_ratio uniform points in a box."""

    def __init__(self, global_sigma=0.5, global_sigma_xyz=None, local_sigma=0.01):
        self.global_sigma_xyz = torch.ones(3) * global_sigma if global_sigma_xyz is None else \
            torch.as_tensor(global_sigma_xyz, dtype=torch.float32)
        self.local_sigma = local_sigma

    def get_points(self, pc_input, local_sigma=None, global_ratio=0.125):
        gs = self.global_sigma_xyz.to(pc_input.device)
        B, N, D = pc_input.shape
        local = pc_input + torch.randn_like(pc_input) * (self.local_sigma if local_sigma is None else local_sigma)
        glob = torch.rand(B, int(N * global_ratio), D, device=pc_input.device) * (gs * 2) - gs
        return torch.cat([local, glob], dim=1)
