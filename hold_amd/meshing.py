"""Canonical meshing on the GPU: ``generate_mesh`` with the reference's signature
(code/src/utils/meshing.py:9-72) on a dense SDF grid + marching tetrahedra (csrc/meshing.hip).

Differences from the reference, by design:
* the reference refines an octree (MISE, code/src/libmise/mise.pyx) from ``res_init`` up ``res_up`` times so that a CPU
  has fewer points to query; here the final-resolution grid ((res_init * 2**res_up + 1)^3 points, 2.1 M for the
  shipped settings) is ONE batched fused-trunk query, so every voxel is evaluated (a superset of what MISE evaluates);
* skimage's Lewiner marching cubes is replaced by marching tetrahedra: the same piecewise-linear level set sampled on
  the same grid edges plus the cube face / body diagonals, different triangulation.  Vertices are welded and faces
  outward-oriented (the reference's ``gradient_direction="ascent"`` + ``faces[:, [0, 2, 1]]``);
* the largest connected component (by area) is kept, as the reference does through trimesh (:62-70), with a label
  propagation on the device.
The returned object quacks like the ``trimesh.Trimesh`` the callers use: ``.vertices``, ``.faces``, ``.area``,
``.export(path)`` (Wavefront OBJ).  If trimesh is importable, a real ``trimesh.Trimesh`` is returned instead.
"""
from __future__ import annotations

import itertools
import os

import numpy as np
import torch

from ._lib import call, ptr

# ---------------------------------------------------------------------------------------------- case tables
_TABLES = None


def _kuhn_tets():
    """6 tetrahedra (c0, c0+e_a, c0+e_a+e_b, c7) for the permutations (a, b, c) of the axes; corner id = x + 2y + 4z."""
    tets = []
    for perm in itertools.permutations((1, 2, 4)):
        tets.append((0, perm[0], perm[0] | perm[1], 7))
    return tets


def tables():
    """(tet_corner [6,4], ntri [6,16], tri_tab [6,16,2,3,2]) int8.  Case bit k of a tet = its k-th corner is inside
    (value < level).  Triangles are wound so that their normal points from inside to outside; the winding is decided
    on a prototype (crossings at edge midpoints) and is constant within a sign case."""
    global _TABLES
    if _TABLES is not None:
        return _TABLES
    tets = _kuhn_tets()
    cpos = np.array([[c & 1, (c >> 1) & 1, (c >> 2) & 1] for c in range(8)], dtype=np.float64)
    tet_corner = np.array(tets, dtype=np.int8)
    ntri = np.zeros((6, 16), dtype=np.int8)
    tri_tab = np.zeros((6, 16, 2, 3, 2), dtype=np.int8)
    for t, tet in enumerate(tets):
        for m in range(1, 15):
            ins = [k for k in range(4) if (m >> k) & 1]
            out = [k for k in range(4) if not (m >> k) & 1]
            if len(ins) == 1:
                tris = [[(ins[0], o) for o in out]]
            elif len(ins) == 3:
                tris = [[(i, out[0]) for i in ins]]
            else:
                a, b = ins
                c, d = out
                quad = [(a, c), (a, d), (b, d), (b, c)]
                tris = [[quad[0], quad[1], quad[2]], [quad[0], quad[2], quad[3]]]
            pin = cpos[[tet[k] for k in ins]].mean(0)
            pout = cpos[[tet[k] for k in out]].mean(0)
            for j, tri in enumerate(tris):
                pts = [0.5 * (cpos[tet[i]] + cpos[tet[o]]) for i, o in tri]
                nrm = np.cross(pts[1] - pts[0], pts[2] - pts[0])
                if np.dot(nrm, pout - pin) < 0:
                    tri = [tri[0], tri[2], tri[1]]
                for v, (i, o) in enumerate(tri):
                    ca, cb = sorted((tet[i], tet[o]))
                    assert ca & cb == ca  # nested corners along a Kuhn path
                    tri_tab[t, m, j, v] = (ca, cb)
            ntri[t, m] = len(tris)
    _TABLES = (tet_corner, ntri, tri_tab)
    return _TABLES


_DEV_TABLES = {}


def _device_tables(dev):
    key = str(dev)
    if key not in _DEV_TABLES:
        _DEV_TABLES[key] = tuple(torch.from_numpy(np.ascontiguousarray(t)).to(dev) for t in tables())
    return _DEV_TABLES[key]


# ---------------------------------------------------------------------------------------------- extraction
def marching_tetrahedra(sdf_grid, origin, spacing, level=0.0):
    """sdf_grid [n,n,n] fp32 CUDA (x-major) -> (verts [V,3] fp32 world coordinates, faces [F,3] int64), welded,
    outward oriented (towards larger values)."""
    n = sdf_grid.shape[0]
    assert sdf_grid.shape == (n, n, n) and sdf_grid.is_cuda
    dev = sdf_grid.device
    sdf = sdf_grid.contiguous().float()
    tet_corner, ntri, tri_tab = _device_tables(dev)
    npts = n * n * n
    edge_flag = torch.empty(npts * 7, dtype=torch.int32, device=dev)
    cube_ntri = torch.empty(npts, dtype=torch.int32, device=dev)
    call("hold_mt_classify", ptr(sdf), n, float(level), ptr(ntri), ptr(tet_corner), ptr(edge_flag), ptr(cube_ntri))
    edge_inc = torch.cumsum(edge_flag, 0, dtype=torch.int64)
    cube_inc = torch.cumsum(cube_ntri, 0, dtype=torch.int64)
    V, F = int(edge_inc[-1]), int(cube_inc[-1])
    edge_scan = (edge_inc - edge_flag).contiguous()
    cube_scan = (cube_inc - cube_ntri).contiguous()
    verts = torch.empty(V, 3, device=dev)
    faces = torch.empty(F, 3, dtype=torch.int64, device=dev)
    if V == 0 or F == 0:
        return verts, faces
    call("hold_mt_vertices", ptr(sdf), n, float(level), float(origin[0]), float(origin[1]), float(origin[2]),
         float(spacing), ptr(edge_flag), ptr(edge_scan), ptr(verts))
    call("hold_mt_triangles", ptr(sdf), n, float(level), ptr(ntri), ptr(tet_corner), ptr(tri_tab), ptr(cube_ntri),
         ptr(cube_scan), ptr(edge_scan), ptr(faces))
    return verts, faces


def largest_component(verts, faces):
    """keep the connected component with the largest surface area (meshing.py:62-70); re-indexes the vertices."""
    V = verts.shape[0]
    if faces.shape[0] == 0:
        return verts, faces
    lab = torch.arange(V, device=verts.device)
    e = torch.cat([faces[:, [0, 1]], faces[:, [1, 2]], faces[:, [2, 0]]], 0)
    a, b = e[:, 0], e[:, 1]
    for _ in range(10_000):
        new = lab.clone()
        new.scatter_reduce_(0, a, lab[b], reduce="amin")
        new.scatter_reduce_(0, b, lab[a], reduce="amin")
        new = new[new]  # pointer jumping
        if torch.equal(new, lab):
            break
        lab = new
    fl = lab[faces[:, 0]]
    p0, p1, p2 = verts[faces[:, 0]], verts[faces[:, 1]], verts[faces[:, 2]]
    area = 0.5 * torch.linalg.norm(torch.cross(p1 - p0, p2 - p0, dim=1), dim=1)
    tot = torch.zeros(V, device=verts.device, dtype=area.dtype).index_add_(0, fl, area)
    keep_f = fl == torch.argmax(tot)
    faces = faces[keep_f]
    used = torch.zeros(V, dtype=torch.bool, device=verts.device)
    used[faces.reshape(-1)] = True
    remap = torch.cumsum(used.long(), 0) - 1
    return verts[used], remap[faces]


class TriMesh:
    """the slice of ``trimesh.Trimesh`` the reference touches (hold.py:151-167, object_node.py:123-132)."""

    def __init__(self, vertices, faces, vertex_values=None):
        self.vertices = np.asarray(vertices, dtype=np.float64)
        self.faces = np.asarray(faces, dtype=np.int64)
        self.vertex_values = vertex_values

    @property
    def area(self):
        v, f = self.vertices, self.faces
        return float(0.5 * np.linalg.norm(np.cross(v[f[:, 1]] - v[f[:, 0]], v[f[:, 2]] - v[f[:, 0]]), axis=1).sum())

    @property
    def volume(self):
        v, f = self.vertices, self.faces
        return float(np.einsum("ij,ij->i", v[f[:, 0]], np.cross(v[f[:, 1]], v[f[:, 2]])).sum() / 6.0)

    def export(self, path):
        d = os.path.dirname(path)
        if d:
            os.makedirs(d, exist_ok=True)
        with open(path, "w") as fh:
            for p in self.vertices:
                fh.write(f"v {p[0]:.8f} {p[1]:.8f} {p[2]:.8f}\n")
            for t in self.faces:
                fh.write(f"f {t[0] + 1} {t[1] + 1} {t[2] + 1}\n")
        return path


def _wrap(verts, faces):
    v, f = verts.detach().cpu().numpy(), faces.detach().cpu().numpy()
    try:
        import trimesh  # noqa: WPS433 -- optional; absent in this image

        if hasattr(trimesh, "Trimesh") and trimesh.Trimesh is not object:
            return trimesh.Trimesh(v, f, process=False)
    except Exception:
        pass
    return TriMesh(v, f)


def sdf_grid(func, verts, res, device, point_batch=1 << 21, scale=1.1):
    """the padded-bbox grid of generate_mesh (meshing.py:12-19,33-34): -> (values [n,n,n], origin [3], spacing)."""
    verts = np.asarray(verts, dtype=np.float64)
    bmin, bmax = verts.min(axis=0), verts.max(axis=0)
    center = (bmin + bmax) * 0.5
    gt_scale = float((bmax - bmin).max())
    n = res + 1
    spacing = scale * gt_scale / res
    origin = center - 0.5 * scale * gt_scale
    ax = torch.arange(n, device=device, dtype=torch.float32)
    gx, gy, gz = torch.meshgrid(ax, ax, ax, indexing="ij")
    o = torch.tensor(origin, device=device, dtype=torch.float32)
    pts = torch.stack([gx, gy, gz], -1).reshape(-1, 3) * float(spacing) + o
    vals = torch.empty(pts.shape[0], device=device)
    for lo in range(0, pts.shape[0], point_batch):
        out = func(pts[lo:lo + point_batch])
        vals[lo:lo + point_batch] = out["sdf"].reshape(-1).float()
    return vals.view(n, n, n), origin, spacing


def generate_mesh(func, verts, level_set=0, res_init=32, res_up=3, point_batch=5000, device="cuda"):
    """code/src/utils/meshing.py:9-72.  func(points [P,3] cuda) -> {"sdf": [P]}; verts: points spanning the tight bbox.
    ``point_batch`` is accepted for signature parity (the reference queries 5-10 k points at a time; here the whole
    grid goes through in 2 M-point launches)."""
    res = res_init * 2 ** res_up
    vals, origin, spacing = sdf_grid(func, verts, res, torch.device(device))
    v, f = marching_tetrahedra(vals, origin, spacing, float(level_set))
    v, f = largest_component(v, f)
    return _wrap(v, f)
