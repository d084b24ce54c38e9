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


class MeshIndex:
    """Acceleration structure for the per-ray off-surface test against ONE canonical mesh and ONE threshold (built when
    the loss-target mesh changes: every 200 steps for a hand, every canonical re-meshing for the object): exact signed
    distances at the nodes of a uniform grid with h * sqrt(3) < threshold, and per-cell lists of the triangles whose
    threshold-dilated bounding box touches the cell.  See csrc/geometry.hip:ray_off_surface_kernel for how the two
    decide ``min over a ray's samples of sd > threshold`` with the decisions of the brute-force test."""

    def __init__(self, verts, faces, threshold, max_nodes=160):
        assert verts.dim() == 2 and faces.dim() == 2
        dev = verts.device
        self.verts = verts.detach().contiguous().float()
        self.faces = faces.to(torch.int32).contiguous()
        self.thr = float(threshold)
        lo, hi = self.verts.min(0).values, self.verts.max(0).values
        ext = float((hi - lo).max())
        h = 0.5 * self.thr  # h * sqrt(3) = 0.87 thr < thr
        margin = self.thr + 2.0 * h
        G = int(torch.ceil(torch.tensor((ext + 2 * margin) / h))) + 1
        # a mesh too large for the node grid (an object whose canonical extent is >> the threshold) falls back to the
        # exact brute-force test the index accelerates (what the reference always runs through kaolin)
        self.brute = G > max_nodes
        if self.brute:
            return
        self.G, self.h = G, h
        ctr = 0.5 * (lo + hi)
        self.origin = (ctr - 0.5 * (G - 1) * h).tolist()
        ax = torch.arange(G, device=dev, dtype=torch.float32) * h
        gx, gy, gz = torch.meshgrid(ax + self.origin[0], ax + self.origin[1], ax + self.origin[2], indexing="ij")
        nodes = torch.stack([gx, gy, gz], -1).reshape(1, -1, 3)
        self.node_sdf = mesh_sdf(nodes, self.verts, self.faces).reshape(-1).contiguous()
        # cell lists (CSR) from the thr-dilated triangle bounding boxes
        tri = self.verts[self.faces.long()]  # [F,3,3]
        o = torch.tensor(self.origin, device=dev)
        c_lo = torch.floor((tri.min(1).values - self.thr - o) / h).long().clamp_(0, G - 2)
        c_hi = torch.floor((tri.max(1).values + self.thr - o) / h).long().clamp_(0, G - 2)
        span = (c_hi - c_lo + 1)
        D = [int(span[:, k].max()) for k in range(3)]
        F = self.faces.shape[0]
        fid = torch.arange(F, device=dev)
        cells, tris = [], []
        for dx in range(D[0]):
            for dy in range(D[1]):
                for dz in range(D[2]):
                    ok = (dx < span[:, 0]) & (dy < span[:, 1]) & (dz < span[:, 2])
                    if not bool(ok.any()):
                        continue
                    c = ((c_lo[ok, 0] + dx) * (G - 1) + (c_lo[ok, 1] + dy)) * (G - 1) + (c_lo[ok, 2] + dz)
                    cells.append(c)
                    tris.append(fid[ok])
        cells, tris = torch.cat(cells), torch.cat(tris)
        order = torch.argsort(cells, stable=True)
        self.cell_tris = tris[order].to(torch.int32).contiguous()
        counts = torch.bincount(cells, minlength=(G - 1) ** 3)
        self.cell_start = torch.cat([torch.zeros(1, dtype=torch.long, device=dev), torch.cumsum(counts, 0)]).to(torch.int32).contiguous()

    def off_surface(self, x_cano, num_rays):
        """x_cano [..., 3] canonical sample points, ray-major (num_rays x S) -> bool [num_rays]."""
        x = x_cano.detach().reshape(-1, x_cano.shape[-1])
        if self.brute:
            return check_off_in_surface_points_cano_mesh(self.verts, self.faces, x[None, :, :3].float(), num_rays, self.thr)[0]
        if x.dtype != torch.float32 or x.stride(1) != 1:
            x = x.float().contiguous()
        S = x.shape[0] // num_rays
        assert S * num_rays == x.shape[0]
        out = torch.empty(num_rays, dtype=torch.uint8, device=x.device)
        call("hold_ray_off_surface", ptr(x), x.stride(0), num_rays, S, ptr(self.node_sdf), self.G, self.origin[0],
             self.origin[1], self.origin[2], self.h, self.thr, ptr(self.cell_start), ptr(self.cell_tris), ptr(self.verts),
             ptr(self.faces), ptr(out))
        return out.bool()


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
        if self.global_sigma_xyz.device != pc_input.device:  # moved once (every host -> device copy drains the stream)
            self.global_sigma_xyz = self.global_sigma_xyz.to(pc_input.device)
        gs = self.global_sigma_xyz
        B, N, D = pc_input.shape
        local = pc_input + torch.randn_like(pc_input) * (self.local_sigma if local_sigma is None else local_sigma)
        glob = torch.rand(B, int(N * global_ratio), D, device=pc_input.device) * (gs * 2) - gs
        return torch.cat([local, glob], dim=1)


def subdivide_loop(verts, faces):
    """One iteration of Loop subdivision of a triangle mesh (what ``trimesh.remesh.subdivide_loop(v, f, iterations=1)``
    does for hold_utils.subdivide_cano, code/src/hold/hold_utils.py:137-146): every triangle -> 4; edge ("odd")
    vertices at 3/8 (a+b) + 1/8 (c+d) for interior edges, midpoints on boundary edges; original ("even") vertices
    relaxed with Loop's beta(k).  verts [V,3] float, faces [F,3] int64 -> ([V+E,3], [4F,3]); new vertex order =
    even vertices then one vertex per unique edge.  Runs on the tensors' device (a once-per-200-steps host-side op).
    trimesh is not available in this image: the scheme is restated from Loop's rules, parity with trimesh unpinned."""
    V, dev = verts.shape[0], verts.device
    f = faces.long()
    e = torch.stack([f[:, [0, 1]], f[:, [1, 2]], f[:, [2, 0]]], 1).reshape(-1, 2)  # 3 half-edges per face
    opp = torch.stack([f[:, 2], f[:, 0], f[:, 1]], 1).reshape(-1)  # vertex opposite each half-edge
    lo, hi = e.min(1).values, e.max(1).values
    key = lo * V + hi
    uniq, inverse, counts = torch.unique(key, return_inverse=True, return_counts=True)
    E = uniq.shape[0]
    a, b = uniq // V, uniq % V
    opp_sum = torch.zeros(E, 3, device=dev, dtype=verts.dtype).index_add_(0, inverse, verts[opp])
    interior = (counts == 2)[:, None]
    odd = torch.where(interior, 0.375 * (verts[a] + verts[b]) + 0.125 * opp_sum, 0.5 * (verts[a] + verts[b]))
    # even vertices: neighbours through unique edges
    nsum = torch.zeros(V, 3, device=dev, dtype=verts.dtype)
    nsum.index_add_(0, a, verts[b]).index_add_(0, b, verts[a])
    k = torch.zeros(V, device=dev, dtype=verts.dtype)
    one = torch.ones(E, device=dev, dtype=verts.dtype)
    k.index_add_(0, a, one).index_add_(0, b, one)
    kk = k.clamp(min=1)
    beta = (40.0 - (2.0 * torch.cos(2 * torch.pi / kk) + 3) ** 2) / (64 * kk)
    even = beta[:, None] * nsum + (1 - kk * beta)[:, None] * verts
    # boundary vertices: 1/8 (sum of the two boundary neighbours) + 3/4 v
    bmask = counts == 1
    if bool(bmask.any()):
        bs = torch.zeros(V, 3, device=dev, dtype=verts.dtype)
        bs.index_add_(0, a[bmask], verts[b[bmask]]).index_add_(0, b[bmask], verts[a[bmask]])
        on_b = torch.zeros(V, dtype=torch.bool, device=dev)
        on_b[a[bmask]] = True
        on_b[b[bmask]] = True
        even = torch.where(on_b[:, None], 0.125 * bs + 0.75 * verts, even)
    oi = inverse.reshape(-1, 3) + V  # odd vertex on edges (01, 12, 20) of each face
    nf = torch.stack([f[:, 0], oi[:, 0], oi[:, 2], oi[:, 0], f[:, 1], oi[:, 1], oi[:, 2], oi[:, 1], f[:, 2],
                      oi[:, 0], oi[:, 1], oi[:, 2]], 1).reshape(-1, 3)
    return torch.cat([even, odd], 0), nf
