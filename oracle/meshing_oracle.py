"""TEST INFRASTRUCTURE ONLY -- CPU (numpy, fp64) restatement of the iso-surface extraction behind
``generate_mesh`` (code/src/utils/meshing.py:9-72).

The reference extracts the zero level set of the canonical SDF with MISE (code/src/libmise/mise.pyx) + skimage's
Lewiner marching cubes (scikit-image is not installed here and not vendored in /root/reference: **parity with
skimage's triangulation is unpinned**).  Both the reference and hold_amd/meshing.py triangulate the same object -- the
piecewise-linear interpolant of the grid samples -- so this oracle is pinned against closed-form level sets instead
(tests/test_oracle_golden.py: sphere area / volume / vertex residuals) and restates the product's method (marching
tetrahedra on the Kuhn decomposition) *without its case tables*: every tetrahedron is cut from its actual values and
wound by the numeric inside->outside direction, so a wrong table entry in the product shows up as a differing
triangle set.
"""
from __future__ import annotations

import itertools

import numpy as np


def kuhn_tets():
    return [(0, p[0], p[0] | p[1], 7) for p in itertools.permutations((1, 2, 4))]


def marching_tetrahedra(vals, origin, spacing, level=0.0):
    """vals [n,n,n] (x-major) -> list of triangles, each a (3,3) array of world-space points, outward wound."""
    n = vals.shape[0]
    origin = np.asarray(origin, dtype=np.float64)
    tris = []
    corner = np.array([[c & 1, (c >> 1) & 1, (c >> 2) & 1] for c in range(8)])
    inside = vals < level
    # cubes whose 8 corners are not all on one side
    s = np.zeros((n - 1, n - 1, n - 1), dtype=np.int32)
    for c in corner:
        s += inside[c[0]:n - 1 + c[0], c[1]:n - 1 + c[1], c[2]:n - 1 + c[2]]
    for ix, iy, iz in np.argwhere((s > 0) & (s < 8)):
        base = np.array([ix, iy, iz])
        for tet in kuhn_tets():
            P = np.array([base + corner[c] for c in tet], dtype=np.float64)
            v = np.array([vals[tuple(base + corner[c])] for c in tet], dtype=np.float64)
            ins = [k for k in range(4) if v[k] < level]
            out = [k for k in range(4) if not v[k] < level]
            if not ins or not out:
                continue

            def cut(i, o):
                # interpolate from the lattice-lower end of the edge, exactly like a shared-edge implementation must
                a, b = (i, o) if tuple(P[i]) < tuple(P[o]) else (o, i)
                t = (level - v[a]) / (v[b] - v[a])
                return P[a] + t * (P[b] - P[a])

            if len(ins) == 1:
                polys = [[cut(ins[0], o) for o in out]]
            elif len(ins) == 3:
                polys = [[cut(i, out[0]) for i in ins]]
            else:
                a, b = ins
                c, d = out
                q = [cut(a, c), cut(a, d), cut(b, d), cut(b, c)]
                polys = [[q[0], q[1], q[2]], [q[0], q[2], q[3]]]
            dirn = P[out].mean(0) - P[ins].mean(0)
            for tri in polys:
                tri = np.array(tri)
                nrm = np.cross(tri[1] - tri[0], tri[2] - tri[0])
                if np.dot(nrm, dirn) < 0:  # wind so that the normal points from inside to outside
                    tri = tri[[0, 2, 1]]
                tris.append(origin + spacing * tri)
    return tris


def canonical_triangles(tris, decimals=5):
    """orientation-preserving canonical form: rotate each triangle so its smallest vertex comes first -> sorted list of
    tuples.  Degenerate (zero-area) triangles are dropped: their winding is not defined."""
    out = []
    for t in tris:
        t = np.asarray(t, dtype=np.float64)
        if np.linalg.norm(np.cross(t[1] - t[0], t[2] - t[0])) < 1e-9:
            continue
        r = np.round(t, decimals) + 0.0
        keys = [tuple(p) for p in r]
        k = keys.index(min(keys))
        out.append(tuple(keys[(k + i) % 3] for i in range(3)))
    return sorted(out)


def mesh_stats(verts, faces):
    v, f = np.asarray(verts, dtype=np.float64), np.asarray(faces)
    a, b, c = v[f[:, 0]], v[f[:, 1]], v[f[:, 2]]
    area = 0.5 * np.linalg.norm(np.cross(b - a, c - a), axis=1).sum()
    vol = np.einsum("ij,ij->i", a, np.cross(b, c)).sum() / 6.0
    e = np.concatenate([f[:, [0, 1]], f[:, [1, 2]], f[:, [2, 0]]])
    key = e[:, 0].astype(np.int64) * (v.shape[0] + 1) + e[:, 1]
    rkey = e[:, 1].astype(np.int64) * (v.shape[0] + 1) + e[:, 0]
    closed = len(np.unique(key)) == len(key) and set(key.tolist()) == set(rkey.tolist())
    return dict(area=float(area), volume=float(vol), closed_oriented=bool(closed))


def triangles_match(tris_a, tris_b, atol=2e-5):
    """bijection between two triangle soups up to cyclic rotation (orientation preserved) and `atol` per coordinate --
    what a comparison of an fp32 extraction with the fp64 oracle needs (rounded keys flip at decimal boundaries)."""
    from scipy.spatial import cKDTree
    a = np.asarray([t for t in tris_a if np.linalg.norm(np.cross(t[1] - t[0], t[2] - t[0])) > 1e-9], dtype=np.float64)
    b = np.asarray([t for t in tris_b if np.linalg.norm(np.cross(t[1] - t[0], t[2] - t[0])) > 1e-9], dtype=np.float64)
    if len(a) != len(b):
        return False, f"{len(a)} vs {len(b)} non-degenerate triangles"
    tree = cKDTree(b.mean(1))
    used = np.zeros(len(b), dtype=bool)
    for i, t in enumerate(a):
        hit = False
        for j in tree.query_ball_point(t.mean(0), r=4 * atol):
            if used[j]:
                continue
            if any(np.abs(t - np.roll(b[j], r, axis=0)).max() < atol for r in range(3)):
                used[j] = hit = True
                break
        if not hit:
            return False, f"triangle {i} of the first set has no partner: {t.tolist()}"
    return True, ""
