"""TEST INFRASTRUCTURE ONLY -- CPU restatement (fp64-capable torch) of the training loss-target geometry of
code/src/engine/volsdf_utils.py:172-217 and code/src/hold/hold_utils.py:22-55,274-303.

The reference gets point->mesh distance and the inside test from kaolin v0.10
(metrics.trianglemesh.point_to_mesh_distance, ops.mesh.check_sign), which is not available here and is not vendored in
/root/reference: **parity against kaolin is unpinned**.  Both quantities are, however, defined by exact geometry for the
watertight meshes the reference feeds them (the sealed, once-subdivided canonical MANO, mano_node.py:126-135):
  * unsigned distance = min over faces of the Euclidean distance to the closest point of the triangle
    (region classification of Ericson, Real-Time Collision Detection 5.1.5);
  * inside <=> |generalised winding number| > 1/2 (sum of the signed solid angles of the faces, Van Oosterom &
    Strackee 1983); kaolin's ray-parity test agrees with it for every point that is not on the surface.
So this restatement is pinned against closed-form SDFs instead (tests/test_oracle_golden.py: box, tetrahedron).
"""
from __future__ import annotations

import math

import torch


def closest_point_d2(p, a, b, c):
    """squared distance from points p [...,3] to triangles (a, b, c) [...,3] (broadcast)."""
    ab, ac, ap = b - a, c - a, p - a
    d1, d2 = (ab * ap).sum(-1), (ac * ap).sum(-1)
    bp = p - b
    d3, d4 = (ab * bp).sum(-1), (ac * bp).sum(-1)
    cp = p - c
    d5, d6 = (ab * cp).sum(-1), (ac * cp).sum(-1)
    vc = d1 * d4 - d3 * d2
    vb = d5 * d2 - d1 * d6
    va = d3 * d6 - d5 * d4
    tiny = 1e-30
    # candidates
    q_a, q_b, q_c = a, b, c
    t_ab = d1 / (d1 - d3).clamp_min(tiny)
    q_ab = a + ab * t_ab[..., None]
    t_ac = d2 / (d2 - d6).clamp_min(tiny)
    q_ac = a + ac * t_ac[..., None]
    t_bc = (d4 - d3) / ((d4 - d3) + (d5 - d6)).clamp_min(tiny)
    q_bc = b + (c - b) * t_bc[..., None]
    den = (va + vb + vc)
    den = torch.where(den.abs() < tiny, torch.full_like(den, tiny), den)
    v, w = vb / den, vc / den
    q_in = a + ab * v[..., None] + ac * w[..., None]
    r_a = (d1 <= 0) & (d2 <= 0)
    r_b = (d3 >= 0) & (d4 <= d3)
    r_ab = (vc <= 0) & (d1 >= 0) & (d3 <= 0)
    r_c = (d6 >= 0) & (d5 <= d6)
    r_ac = (vb <= 0) & (d2 >= 0) & (d6 <= 0)
    r_bc = (va <= 0) & ((d4 - d3) >= 0) & ((d5 - d6) >= 0)
    q = q_in
    for cond, cand in ((r_bc, q_bc), (r_ac, q_ac), (r_c, q_c), (r_ab, q_ab), (r_b, q_b), (r_a, q_a)):  # first match wins
        q = torch.where(cond[..., None], cand, q)
    return ((p - q) ** 2).sum(-1)


def solid_angle(p, a, b, c):
    """signed solid angle of triangle (a, b, c) seen from p."""
    ra, rb, rc = a - p, b - p, c - p
    la, lb, lc = ra.norm(dim=-1), rb.norm(dim=-1), rc.norm(dim=-1)
    num = (ra * torch.cross(rb, rc, dim=-1)).sum(-1)
    den = la * lb * lc + (ra * rb).sum(-1) * lc + (rb * rc).sum(-1) * la + (rc * ra).sum(-1) * lb
    return 2.0 * torch.atan2(num, den)


def mesh_sdf(points, verts, faces, chunk=4096):
    """signed distance (negative inside) of points [P,3] to the closed triangle mesh (verts [V,3], faces [F,3])."""
    tri = verts[faces]  # [F,3,3]
    out = []
    for lo in range(0, points.shape[0], chunk):
        p = points[lo:lo + chunk, None, :]
        a, b, c = tri[None, :, 0], tri[None, :, 1], tri[None, :, 2]
        d2 = closest_point_d2(p, a, b, c).min(dim=1).values
        om = solid_angle(p, a, b, c).sum(dim=1)
        sign = torch.where(om.abs() > 2 * math.pi, -torch.ones_like(d2), torch.ones_like(d2))
        out.append(sign * d2.sqrt())
    return torch.cat(out)


def compute_mano_cano_sdf(mesh_v, mesh_f, x_cano):
    """volsdf_utils.py:172-186: mesh_v [B,V,3], x_cano [B,P,3] -> [B,P]."""
    return torch.stack([mesh_sdf(x_cano[b], mesh_v[b], mesh_f) for b in range(x_cano.shape[0])])


def check_off_in_surface_points_cano_mesh(mesh_v, mesh_f, x_cano, num_pixels_total, threshold=0.05):
    """volsdf_utils.py:189-217: per ray, min over its samples of the signed distance -> (off-surface, in-surface) masks."""
    sd = compute_mano_cano_sdf(mesh_v, mesh_f, x_cano).reshape(num_pixels_total, -1, 1)
    minimum = sd.min(dim=1).values
    return (minimum > threshold).squeeze(1), (minimum <= 0.0).squeeze(1)


def box_mesh(h):
    """closed, outward-oriented triangle mesh of the axis-aligned box [-h, h] (h: 3 half extents)."""
    hx, hy, hz = h
    v = torch.tensor([[-hx, -hy, -hz], [hx, -hy, -hz], [hx, hy, -hz], [-hx, hy, -hz], [-hx, -hy, hz], [hx, -hy, hz],
                      [hx, hy, hz], [-hx, hy, hz]], dtype=torch.float64)
    f = torch.tensor([[0, 2, 1], [0, 3, 2], [4, 5, 6], [4, 6, 7], [0, 1, 5], [0, 5, 4], [2, 3, 7], [2, 7, 6],
                      [1, 2, 6], [1, 6, 5], [3, 0, 4], [3, 4, 7]])
    return v, f


def box_sdf(p, h):
    """closed-form signed distance to the same box."""
    q = p.abs() - torch.as_tensor(h, dtype=p.dtype)
    return q.clamp_min(0).norm(dim=-1) + q.max(dim=-1).values.clamp_max(0)
