"""TEST INFRASTRUCTURE ONLY -- CPU (numpy, fp64) restatement of the iso-surface extraction behind
``generate_mesh`` (code/src/utils/meshing.py:9-72).

The reference extracts the zero level set of the canonical SDF with MISE (code/src/libmise/mise.pyx) + skimage's
Lewiner marching cubes (scikit-image is not installed here and not vendored in /root/reference: **parity with
skimage's triangulation is unpinned**).  The MISE refinement itself IS pinned: `MISE` below restates it and
tests/test_meshing_cpu.py compares it, query by query and in the final dense grid, with the reference's own module
compiled by oracle/build_ref.py (oracle/_ref/mise.so).  Both the reference and hold_amd/meshing.py triangulate the same object -- the
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


# ---------------------------------------------------------------------------------------------------------------------
# Marching CUBES, restated without case tables (the triangulation family of the reference: skimage's
# ``marching_cubes_lewiner`` behind code/src/utils/meshing.py:48-50).  Every cube with a sign change is polygonised from
# its actual values: the level-set crossings on its 12 edges are joined across each of the 6 faces (a face with two
# crossings: one segment; with four -- the ambiguous face -- the pairing is chosen by the sign of the bilinear saddle
# value, Lewiner's / Nielson-Hamann's asymptotic decider, which depends on the face's four values only and is therefore
# the same in both cubes sharing the face: the surface is watertight), the segments close into loops, and each loop is
# fanned from its centroid.  Interior ("tunnel") ambiguities, which Lewiner's tables also resolve, are not: they need
# a cube whose face pairings admit two topologies, which a distance-like field sampled finer than its features does not
# produce.  scikit-image itself is not installed here: the RESTATEMENT is pinned to closed-form level sets only.
_CUBE_EDGES = [(0, 1), (2, 3), (4, 5), (6, 7), (0, 2), (1, 3), (4, 6), (5, 7), (0, 4), (1, 5), (2, 6), (3, 7)]  # corner id = x + 2y + 4z
# faces as corner cycles (consecutive corners share a cube edge)
_CUBE_FACES = [(0, 2, 6, 4), (1, 3, 7, 5), (0, 1, 5, 4), (2, 3, 7, 6), (0, 1, 3, 2), (4, 5, 7, 6)]
_EDGE_ID = {frozenset(e): i for i, e in enumerate(_CUBE_EDGES)}


def marching_cubes(vals, origin, spacing, level=0.0):
    """vals [n,n,n] (x-major) -> (tris [T,3,3] world-space, outward wound from inside (< level) to outside)."""
    n = vals.shape[0]
    origin = np.asarray(origin, dtype=np.float64)
    corner = np.array([[c & 1, (c >> 1) & 1, (c >> 2) & 1] for c in range(8)])
    inside = vals < level
    s = np.zeros((n - 1, n - 1, n - 1), dtype=np.int32)
    for c in corner:
        s += inside[c[0]:n - 1 + c[0], c[1]:n - 1 + c[1], c[2]:n - 1 + c[2]]
    tris = []
    for ix, iy, iz in np.argwhere((s > 0) & (s < 8)):
        base = np.array([ix, iy, iz])
        v = np.array([vals[tuple(base + corner[c])] for c in range(8)], dtype=np.float64) - level
        pts = {}
        for e, (a, b) in enumerate(_CUBE_EDGES):
            if (v[a] < 0) != (v[b] < 0):
                t = v[a] / (v[a] - v[b])
                pts[e] = base + corner[a] + t * (corner[b] - corner[a])
        nbr = {e: [] for e in pts}
        for face in _CUBE_FACES:
            es = [_EDGE_ID[frozenset((face[k], face[(k + 1) % 4]))] for k in range(4)]
            cr = [k for k in range(4) if es[k] in pts]
            if len(cr) == 2:
                pairs = [(es[cr[0]], es[cr[1]])]
            elif len(cr) == 4:
                f0, f1, f2, f3 = (v[c] for c in face)  # cyclic: f0, f2 and f1, f3 are the diagonals
                # bilinear saddle value S = (f0 f2 - f1 f3) / (f0 + f2 - f1 - f3); on an ambiguous face the denominator has
                # the sign of f0 (= f2), so S has the sign of f0 iff f0 f2 - f1 f3 > 0: then f0 and f2 are connected through
                # the face centre and the crossings pair up AROUND f1 and around f3; otherwise around f0 and f2
                sep_odd = f0 * f2 - f1 * f3 > 0
                pairs = [(es[0], es[1]), (es[2], es[3])] if sep_odd else [(es[3], es[0]), (es[1], es[2])]
            else:
                continue
            for a, b in pairs:
                nbr[a].append(b)
                nbr[b].append(a)
        todo = set(pts)
        while todo:
            start = todo.pop()
            loop, prev, cur = [start], None, start
            while True:
                nxt = [x for x in nbr[cur] if x != prev]
                nx = nxt[0] if nxt else None
                if nx is None or nx == start:
                    break
                loop.append(nx)
                todo.discard(nx)
                prev, cur = cur, nx
            if len(loop) < 3:
                continue
            P = np.array([pts[e] for e in loop], dtype=np.float64)
            ctr = P.mean(0)
            ins_c = corner[[c for c in range(8) if v[c] < 0]].mean(0) + base
            out_c = corner[[c for c in range(8) if not v[c] < 0]].mean(0) + base
            fan = [(ctr, P[k], P[(k + 1) % len(P)]) for k in range(len(P))] if len(P) > 3 else [(P[0], P[1], P[2])]
            nsum = sum(np.cross(b - a, c - a) for a, b, c in fan)
            flip = np.dot(nsum, out_c - ins_c) < 0
            for a, b, c in fan:
                tris.append(np.array([a, c, b] if flip else [a, b, c]) * spacing + origin)
    return np.array(tris).reshape(-1, 3, 3)


def mesh_area_volume(tris):
    """surface area and enclosed volume (divergence theorem) of an outward-wound triangle soup [T,3,3]"""
    a, b, c = tris[:, 0], tris[:, 1], tris[:, 2]
    area = 0.5 * np.linalg.norm(np.cross(b - a, c - a), axis=1).sum()
    vol = np.einsum("ij,ij->i", a, np.cross(b, c)).sum() / 6.0
    return float(area), float(vol)


def point_to_tris(pts, tris, chunk=256):
    """unsigned distance of points [N,3] to a triangle soup [T,3,3] (brute force, exact: Ericson's closest point)"""
    a, b, c = tris[:, 0][None], tris[:, 1][None], tris[:, 2][None]
    ab, ac = b - a, c - a
    out = np.empty(len(pts))
    for i0 in range(0, len(pts), chunk):
        p = pts[i0:i0 + chunk, None, :]
        ap = p - a
        d1, d2 = (ab * ap).sum(-1), (ac * ap).sum(-1)
        bp = p - b
        d3, d4 = (ab * bp).sum(-1), (ac * bp).sum(-1)
        cp = p - c
        d5, d6 = (ab * cp).sum(-1), (ac * cp).sum(-1)
        vc, vb, va = d1 * d4 - d3 * d2, d5 * d2 - d1 * d6, d3 * d6 - d5 * d4
        den = va + vb + vc
        den = np.where(np.abs(den) < 1e-30, 1e-30, den)
        vv, ww = vb / den, vc / den
        q = a + ab * vv[..., None] + ac * ww[..., None]  # interior
        def sel(cond, val, q):
            return np.where(cond[..., None], val, q)
        t_ab = np.clip(d1 / np.where(np.abs(d1 - d3) < 1e-30, 1e-30, d1 - d3), 0, 1)
        t_ac = np.clip(d2 / np.where(np.abs(d2 - d6) < 1e-30, 1e-30, d2 - d6), 0, 1)
        t_bc = np.clip((d4 - d3) / np.where(np.abs((d4 - d3) + (d5 - d6)) < 1e-30, 1e-30, (d4 - d3) + (d5 - d6)), 0, 1)
        q = sel((va <= 0) & (d4 - d3 >= 0) & (d5 - d6 >= 0), b + (c - b) * t_bc[..., None], q)
        q = sel((vb <= 0) & (d2 >= 0) & (d6 <= 0), a + ac * t_ac[..., None], q)
        q = sel((vc <= 0) & (d1 >= 0) & (d3 <= 0), a + ab * t_ab[..., None], q)
        q = sel((d6 >= 0) & (d5 <= d6), np.broadcast_to(c, q.shape), q)
        q = sel((d3 >= 0) & (d4 <= d3), np.broadcast_to(b, q.shape), q)
        q = sel((d1 <= 0) & (d2 <= 0), np.broadcast_to(a, q.shape), q)
        out[i0:i0 + chunk] = np.sqrt(((p - q) ** 2).sum(-1)).min(1)
    return out


# ------------------------------------------------------------------------------------------ MISE (round 4)
class MISE:
    """code/src/libmise/mise.pyx:37-370 restated on dense numpy arrays: an octree over [0, R]^3 (R = resolution_0 << depth)
    whose leaves are refined wherever their 8 corners straddle `threshold`.  Same interface as the reference's extension
    type (query / update / to_dense, .resolution); the order of the points `query()` returns is NOT part of the contract
    (the reference's is the insertion order of a std::vector; callers only pair it with `update`).

      __cinit__ (:46-87)       resolution_0^3 leaf voxels of size 1 << depth, the (resolution_0 + 1)^3 lattice points unknown
      query (:105-127)         the lattice points without a value
      update (:89-103)         store values, then subdivide_voxels (:186-233): every KNOWN point marks the up to 8 LEAF voxels
                               around it (offsets -1 / 0 per axis, :200-216) `next_to_positive` if value >= threshold and
                               `next_to_negative` if value <= threshold; a leaf above the finest level with both marks is
                               split into 8 (:235-283) and the 27 points of its 3 x 3 x 3 lattice are created if missing
      to_dense (:129-163)      values at the known points; the rest forward-filled along x, then y, then z
    """

    def __init__(self, resolution_0, depth, threshold):
        self.resolution_0, self.depth, self.threshold = int(resolution_0), int(depth), float(threshold)
        self.voxel_size_0 = 1 << self.depth
        self.resolution = R = self.resolution_0 * self.voxel_size_0
        self.leaf_level = np.zeros((R, R, R), dtype=np.int8)  # level of the leaf voxel that covers each finest cell
        self.value = np.full((R + 1,) * 3, np.nan)
        self.known = np.zeros((R + 1,) * 3, dtype=bool)
        self.exists = np.zeros((R + 1,) * 3, dtype=bool)
        lat = np.arange(0, R + 1, self.voxel_size_0)
        self.exists[np.ix_(lat, lat, lat)] = True

    def query(self):
        return np.argwhere(self.exists & ~self.known).astype(np.int64)

    def update(self, points, values):
        points = np.asarray(points)
        assert points.shape[0] == np.asarray(values).shape[0] and points.shape[1] == 3
        idx = tuple(points.T)
        if not np.all(self.exists[idx]):
            raise ValueError("Point not in grid!")
        self.value[idx] = values
        self.known[idx] = True
        self._subdivide_voxels()

    def _subdivide_voxels(self):
        R, D = self.resolution, self.depth
        pos = [np.zeros((self.resolution_0 << l,) * 3, dtype=bool) for l in range(D + 1)]
        neg = [np.zeros((self.resolution_0 << l,) * 3, dtype=bool) for l in range(D + 1)]
        pts = np.argwhere(self.known)
        val = self.value[self.known]
        for off in itertools.product((-1, 0), repeat=3):
            adj = pts + np.array(off)
            ok = np.all((adj >= 0) & (adj < R), axis=1)
            a, v = adj[ok], val[ok]
            lev = self.leaf_level[tuple(a.T)]
            for l in range(D + 1):
                m = lev == l
                c = a[m] >> (D - l)
                pos[l][tuple(c[v[m] >= self.threshold].T)] = True
                neg[l][tuple(c[v[m] <= self.threshold].T)] = True
        for l in range(D):  # leaves of the finest level are never split
            size = self.voxel_size_0 >> l
            half = size >> 1
            for c in np.argwhere(pos[l] & neg[l]):
                lo = c * size
                if self.leaf_level[tuple(lo)] != l:  # (marks only ever land on leaves; kept as the reference's is_leaf test)
                    continue
                self.leaf_level[lo[0]:lo[0] + size, lo[1]:lo[1] + size, lo[2]:lo[2] + size] = l + 1
                g = [lo[k] + half * np.arange(3) for k in range(3)]
                self.exists[np.ix_(*g)] = True

    def to_dense(self):
        out = np.where(self.known, self.value, np.nan)
        for axis in range(3):
            for i in range(1, self.resolution + 1):
                cur = out.take(i, axis=axis)
                prev = out.take(i - 1, axis=axis)
                fill = np.where(np.isnan(cur), prev, cur)
                sl = [slice(None)] * 3
                sl[axis] = i
                out[tuple(sl)] = fill
        assert not np.isnan(out).any()
        return out


def mise_dense_grid(func, resolution_0, depth, threshold=0.0):
    """the query loop of generate_mesh (code/src/utils/meshing.py:20-49) around an extractor: func(points [n,3] int64 grid
    coordinates) -> values [n]; returns (dense grid [R+1]^3, number of points evaluated, the extractor)"""
    ext = MISE(resolution_0, depth, threshold)
    n = 0
    pts = ext.query()
    while pts.shape[0] != 0:
        ext.update(pts, np.asarray(func(pts), dtype=np.float64))
        n += pts.shape[0]
        pts = ext.query()
    return ext.to_dense(), n, ext
