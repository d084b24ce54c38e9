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
