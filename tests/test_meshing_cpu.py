"""CPU checks of the canonical-meshing pieces that need no GPU: the marching-tetrahedra case
tables the kernels read (hold_amd/meshing.py) against the table-free oracle, the oracle itself against closed-form
level sets, and the Loop-subdivision restatement against the torch implementation the product uses."""
import math

import numpy as np
import torch

from parity_common import ROOT  # noqa: F401  (sys.path)


def _emulate_kernel(vals, origin, h, level=0.0):
    """numpy walk through the tables exactly as mt_triangles / mt_vertices index them"""
    from hold_amd.meshing import tables
    tet_corner, ntri, tri_tab = [t.astype(int) for t in tables()]
    n = vals.shape[0]
    tris = []
    for ix in range(n - 1):
        for iy in range(n - 1):
            for iz in range(n - 1):
                mask = 0
                for c in range(8):
                    if vals[ix + (c & 1), iy + ((c >> 1) & 1), iz + ((c >> 2) & 1)] < level:
                        mask |= 1 << c
                if mask in (0, 255):
                    continue
                for t in range(6):
                    m = 0
                    for k in range(4):
                        m |= ((mask >> int(tet_corner[t, k])) & 1) << k
                    for j in range(ntri[t, m]):
                        tri = []
                        for v in range(3):
                            ca, cb = [int(q) for q in tri_tab[t, m, j, v]]
                            assert ca & cb == ca and ca != cb
                            pa = np.array([ix + (ca & 1), iy + ((ca >> 1) & 1), iz + ((ca >> 2) & 1)], float)
                            pb = np.array([ix + (cb & 1), iy + ((cb >> 1) & 1), iz + ((cb >> 2) & 1)], float)
                            v0, v1 = vals[tuple(pa.astype(int))], vals[tuple(pb.astype(int))]
                            tri.append(origin + h * (pa + (level - v0) / (v1 - v0) * (pb - pa)))
                        tris.append(np.array(tri))
    return tris


def test_case_tables_reproduce_the_table_free_oracle():
    from oracle import meshing_oracle as mo
    rs = np.random.RandomState(0)
    n = 8
    ax = np.linspace(-1, 1, n)
    X, Y, Z = np.meshgrid(ax, ax, ax, indexing="ij")
    fields = {"sphere": np.sqrt(X ** 2 + Y ** 2 + Z ** 2) - 0.63, "noise": rs.randn(n, n, n),
              "torus": (np.sqrt(X ** 2 + Y ** 2) - 0.5) ** 2 + Z ** 2 - 0.09}
    for name, vals in fields.items():
        a = mo.canonical_triangles(_emulate_kernel(vals, np.array([-1.0, -1, -1]), 2 / (n - 1)))
        b = mo.canonical_triangles(mo.marching_tetrahedra(vals, [-1, -1, -1], 2 / (n - 1)))
        assert len(a) > 100 and a == b, name


def test_meshing_oracle_against_closed_form_sphere():
    from oracle import meshing_oracle as mo
    n, r = 25, 0.7
    ax = np.linspace(-1, 1, n)
    X, Y, Z = np.meshgrid(ax, ax, ax, indexing="ij")
    tris = np.array(mo.marching_tetrahedra(np.sqrt(X ** 2 + Y ** 2 + Z ** 2) - r, [-1, -1, -1], 2 / (n - 1)))
    # weld by coordinates to get an indexed mesh
    pts, inv = np.unique(np.round(tris.reshape(-1, 3), 9), axis=0, return_inverse=True)
    faces = inv.reshape(-1, 3)
    st = mo.mesh_stats(pts, faces)
    assert st["closed_oriented"]
    assert abs(st["volume"] / (4 / 3 * math.pi * r ** 3) - 1) < 0.02 and abs(st["area"] / (4 * math.pi * r ** 2) - 1) < 0.02
    assert np.abs(np.linalg.norm(pts, axis=1) - r).max() < 0.5 * (2 / (n - 1)) ** 2 / r + 1e-9  # chord error of linear interpolation


def test_loop_subdivision_torch_equals_numpy_restatement():
    from hold_amd import fitting as ft, geometry as geo, synthetic as syn
    from oracle import targets_oracle as to
    m = syn.make_mano_model(True)
    v = torch.tensor(m["v_template"], dtype=torch.float64)[None]
    vs, fs = ft.seal_mano_mesh(v, torch.tensor(m["f"]), True)
    vd, fd = geo.subdivide_loop(vs[0], fs)
    ov, of = to.subdivide_loop(vs[0].numpy(), fs.numpy())
    assert vd.shape == (3110, 3) and fd.shape == (6216, 3)  # mano_node.py:126-135
    assert to.mesh_as_triangle_set(vd.numpy(), fd.numpy(), 9) == to.mesh_as_triangle_set(ov, of, 9)
    from oracle import meshing_oracle as mo
    assert mo.mesh_stats(vd.numpy(), fd.numpy())["closed_oriented"]


def _grid(n, lim=1.0):
    ax = np.linspace(-lim, lim, n)
    X, Y, Z = np.meshgrid(ax, ax, ax, indexing="ij")
    return X, Y, Z, ax[1] - ax[0]


def _watertight(tris):
    from collections import Counter
    key = lambda p: tuple(np.round(p, 9))
    cnt = Counter()
    for t in tris:
        for k in range(3):
            cnt[frozenset((key(t[k]), key(t[(k + 1) % 3])))] += 1
    return all(v == 2 for v in cnt.values())


def test_marching_cubes_restatement_is_pinned_to_closed_forms():
    """the table-free marching cubes of oracle/meshing_oracle.py (the triangulation family of the reference's
    skimage call, code/src/utils/meshing.py:48-50): watertight, outward wound, area / volume of a sphere and of a torus
    to the discretisation error of the grid, vertices on the level set of the trilinear interpolant's edges."""
    from oracle import meshing_oracle as mo
    X, Y, Z, h = _grid(33)
    r = 0.63
    mc = mo.marching_cubes(np.sqrt(X ** 2 + Y ** 2 + Z ** 2) - r, [-1, -1, -1], h)
    assert _watertight(mc)
    A, V = mo.mesh_area_volume(mc)
    # an inscribed polyhedron: area and volume fall short of the sphere's by O((h / r)^2) = 0.3 % / 0.6 % on this grid
    assert abs(A / (4 * math.pi * r ** 2) - 1) < 5e-3 and abs(V / (4 / 3 * math.pi * r ** 3) - 1) < 1e-2 and V > 0
    assert np.abs(np.linalg.norm(mc.reshape(-1, 3), axis=1) - r).max() < 0.02 * h + 2e-3  # chord error of a linear edge cut
    R, a = 0.6, 0.22
    mc = mo.marching_cubes(np.sqrt((np.sqrt(X ** 2 + Y ** 2) - R) ** 2 + Z ** 2) - a, [-1, -1, -1], h)
    assert _watertight(mc)
    A, V = mo.mesh_area_volume(mc)
    assert abs(A / (4 * math.pi ** 2 * R * a) - 1) < 1e-2 and abs(V / (2 * math.pi ** 2 * R * a ** 2) - 1) < 2e-2
    # ambiguous faces: a field built to have them (two blobs touching diagonally) still gives a closed surface
    f = np.minimum(np.sqrt((X - 0.31) ** 2 + (Y - 0.31) ** 2 + Z ** 2), np.sqrt((X + 0.31) ** 2 + (Y + 0.31) ** 2 + Z ** 2)) - 0.43
    assert _watertight(mo.marching_cubes(f, [-1, -1, -1], _grid(17)[3] * 0 + h))


def test_marching_tetrahedra_stays_within_half_a_voxel_of_marching_cubes():
    """the product triangulates with marching tetrahedra where the reference uses (Lewiner) marching
    cubes.  Both cut the same grid edges at the same points; MT adds vertices on face / body diagonals.  Stated tolerance
    for downstream consumers (optimize_ckpt.py reads the exported .obj): every MT vertex lies within 0.5 voxel of the MC
    surface and vice versa, area and volume agree to 1 % -- on hand-like and object-like canonical SDFs at the voxel
    sizes the reference meshes with (features >= 3 voxels, smooth fields: at a sharp crease the bound is reached)."""
    from oracle import meshing_oracle as mo
    X, Y, Z, h = _grid(37)
    # "hand": a flat palm ellipsoid with five finger capsules (smooth union); "object": a rounded box with a dent
    def capsule(ax_, ay, bx, by, r):
        pa = np.stack([X - ax_, Y - ay, Z], -1)
        ba = np.array([bx - ax_, by - ay, 0.0])
        t = np.clip((pa @ ba) / (ba @ ba), 0, 1)
        return np.linalg.norm(pa - t[..., None] * ba, axis=-1) - r
    palm = (np.sqrt((X / 0.42) ** 2 + ((Y + 0.25) / 0.38) ** 2 + (Z / 0.16) ** 2) - 1) * 0.16
    hand = palm
    for k in range(5):
        ang = -0.5 + 0.25 * k
        hand = np.minimum(hand, capsule(0.3 * math.sin(ang) * 1.2, 0.05, 0.75 * math.sin(ang), 0.1 + 0.6 * math.cos(ang), 0.07))
    # (off the grid lines: a face lying exactly ON grid nodes puts crossings on the nodes themselves -- degenerate triangles)
    q = np.stack([np.abs(X - 0.013) - 0.447, np.abs(Y + 0.007) - 0.303, np.abs(Z - 0.021) - 0.378], -1)
    box = np.linalg.norm(np.maximum(q, 0), axis=-1) + np.minimum(q.max(-1), 0) - 0.08
    dent = -(np.sqrt((X - 0.507) ** 2 + (Y - 0.003) ** 2 + (Z - 0.396) ** 2) - 0.247)
    obj = 0.05 * np.logaddexp(box / 0.05, dent / 0.05)  # smooth intersection: a learnt SDF has no sharp creases
    for name, f in (("hand", hand), ("object", obj)):
        mc = mo.marching_cubes(f, [-1, -1, -1], h)
        mt = np.array(mo.marching_tetrahedra(f, [-1, -1, -1], h))
        assert _watertight(mc), name
        Ac, Vc = mo.mesh_area_volume(mc)
        At, Vt = mo.mesh_area_volume(mt)
        assert abs(At / Ac - 1) < 1e-2 and abs(Vt / Vc - 1) < 1e-2, (name, At / Ac, Vt / Vc)
        d1 = mo.point_to_tris(np.unique(mt.reshape(-1, 3).round(9), axis=0)[::2], mc)
        d2 = mo.point_to_tris(np.unique(mc.reshape(-1, 3).round(9), axis=0)[::2], mt)
        assert d1.max() < 0.5 * h and d2.max() < 0.5 * h, (name, d1.max() / h, d2.max() / h)


# ------------------------------------------------------------------------------------------ MISE
def _mise_sdfs():
    """canonical-SDF stand-ins on the [0, 128]^3 extraction grid of generate_mesh (meshing.py:20-35): a hand-like union of
    blobs, an object-like box with rounded edges, a torus (genus 1)"""
    def grid_to_unit(p):
        return (np.asarray(p, dtype=np.float64) / 128.0 - 0.5) * 1.1

    def blobs(p):
        x = grid_to_unit(p)
        c = np.array([[0.0, 0.0, 0.0], [0.22, 0.05, 0.0], [-0.2, 0.1, 0.05], [0.05, -0.25, 0.1], [0.1, 0.2, -0.15]])
        r = np.array([0.22, 0.1, 0.09, 0.08, 0.11])
        return (np.linalg.norm(x[:, None, :] - c[None], axis=-1) - r[None]).min(1)

    def box(p):
        q = np.abs(grid_to_unit(p)) - np.array([0.25, 0.15, 0.3])
        return np.linalg.norm(np.maximum(q, 0.0), axis=-1) + np.minimum(q.max(-1), 0.0) - 0.04

    def torus(p):
        x = grid_to_unit(p)
        return np.sqrt((np.sqrt(x[:, 0] ** 2 + x[:, 1] ** 2) - 0.3) ** 2 + x[:, 2] ** 2) - 0.09

    return {"blobs": blobs, "box": box, "torus": torus}


def test_mise_restatement_equals_the_compiled_reference_module():
    """oracle/meshing_oracle.py:MISE against the REFERENCE's own code/src/libmise/mise.pyx, compiled where it lies by
    oracle/build_ref.py into oracle/_ref/mise.so: for the two (res_init, res_up) pairs the reference uses (mano_node.py:144-150:
    64 / 1; object_node.py:114-119: 32 / 2), every refinement round queries the same set of grid points and the final dense
    grids are identical."""
    import pytest
    from oracle import build_ref
    from oracle import meshing_oracle as mo
    build_ref.build()
    ref = build_ref.load_mise()
    if ref is None:
        pytest.skip("oracle/_ref/mise.so not built (needs /root/reference + Cython: this container only)")
    for name, f in _mise_sdfs().items():
        for r0, up in ((64, 1), (32, 2)):
            a, b = ref.MISE(r0, up, 0.0), mo.MISE(r0, up, 0.0)
            assert a.resolution == b.resolution == 128
            rounds = 0
            pa, pb = a.query(), b.query()
            while pa.shape[0] != 0 or pb.shape[0] != 0:
                sa = {tuple(p) for p in pa.tolist()}
                assert sa == {tuple(p) for p in pb.tolist()}, (name, r0, up, rounds, len(pa), len(pb))
                a.update(pa, f(pa).astype(np.float64))
                b.update(pb, f(pb).astype(np.float64))
                pa, pb = a.query(), b.query()
                rounds += 1
            assert rounds >= up + 1  # (fine points on a shared face can mark a coarse neighbour in a later round)
            assert np.array_equal(a.to_dense(), b.to_dense()), (name, r0, up)


def test_dense_grid_has_the_level_set_of_the_mise_grid():
    """What hold_amd/meshing.py does instead of MISE: ONE dense 129^3 query.  On SDFs whose
    features are resolved by the coarse lattice the two grids agree at every point MISE evaluated and have the same sign at
    every other point -- every grid cell is cut (or not) identically, so any iso-surface extractor run on either grid returns
    the same surface; MISE only saves evaluations (here 6-9x).  A feature thinner than a coarse voxel that no coarse lattice
    point sees is MISSED by MISE and found by the dense grid: the dense evaluation is the superset."""
    from oracle import meshing_oracle as mo
    ax = np.arange(129)
    full = np.stack(np.meshgrid(ax, ax, ax, indexing="ij"), -1).reshape(-1, 3)
    for name, f in _mise_sdfs().items():
        dense = f(full).reshape(129, 129, 129)
        for r0, up in ((64, 1), (32, 2)):
            grid, n_eval, ext = mo.mise_dense_grid(f, r0, up)
            assert np.array_equal(grid[ext.known], dense[ext.known])
            assert np.array_equal(grid < 0, dense < 0), (name, r0, up)
            assert n_eval < dense.size / 3
    # a 2.5-cell ball centred in a coarse voxel of the 32 / 2 octree (coarse spacing 4 cells): no coarse corner is inside
    def speck(p):
        return np.linalg.norm(np.asarray(p, dtype=np.float64) - np.array([66.0, 66.0, 66.0]), axis=-1) - 1.25
    grid, _, _ = mo.mise_dense_grid(speck, 32, 2)
    assert not (grid < 0).any() and (speck(full) < 0).any()
