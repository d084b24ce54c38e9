"""CPU: hold_amd.field.pack_weights (stacked, few device ops) against the matrix-by-matrix construction of the same
layouts (tests/pack_reference.py) -- every entry bit-identical, for both node kinds and both arithmetics."""
import pytest
import torch

import hold_amd
from hold_amd import field as F

import pack_reference as ref


def _weights(kind, seed):
    g = torch.Generator().manual_seed(seed)
    spec = F.FieldSpec(kind)
    dims = [(256, 84)] + [(256, 256)] * 2 + [(217, 256)] + [(256, 256)] * 4 + [(257, 256)]
    iw = [torch.randn(n, k, generator=g) / k ** 0.5 for n, k in dims]
    ib = [torch.randn(n, generator=g) * 0.1 for n, _ in dims]
    rdims = [(256, spec.rin_dim)] + [(256, 256)] * 3 + [(3, 256)]
    rw = [torch.randn(n, k, generator=g) / k ** 0.5 for n, k in rdims]
    rb = [torch.randn(n, generator=g) * 0.1 for n, _ in rdims]
    return spec, iw, ib, rw, rb


def _same(a, b, path):
    if a is None or b is None:
        return  # RT[4] is no longer built (the colour head has its own kernels)
    if torch.is_tensor(a):
        assert a.shape == b.shape and a.dtype == b.dtype, (path, a.shape, b.shape, a.dtype, b.dtype)
        assert torch.equal(a, b), path
    elif isinstance(a, (list, tuple)):
        assert len(a) == len(b), path
        for i, (x, y) in enumerate(zip(a, b)):
            _same(x, y, f"{path}[{i}]")
    else:
        assert a == b, path


@pytest.mark.parametrize("kind", ["hand", "object"])
@pytest.mark.parametrize("mode", ["f32x6", "f32"])
@pytest.mark.parametrize("with_render", [True, False])
def test_pack_weights_matches_the_matrix_by_matrix_layouts(kind, mode, with_render):
    prev = hold_amd.precision()
    hold_amd.set_precision(mode)
    try:
        spec, iw, ib, rw, rb = _weights(kind, 3)
        args = (spec, iw, ib, rw if with_render else None, rb if with_render else None, True)
        new, old = F.pack_weights(*args), ref.pack_weights(*args)
        # "trunk_r6" (the stream of the register-resident trunk) has no matrix-by-matrix predecessor: its layout is held to
        # a lane-level model of the MFMA register order in tests/test_r6_pack_cpu.py
        r6_only = {"trunk_r6", "chain_bwd_r6", "w8_feat_r6", "R_r6", "RT_r6"}
        assert set(new) - r6_only == set(old)
        for k in old:
            _same(new[k], old[k], k)
        if mode == "f32x6":  # hold_gemm_r6 packs (one gather for all of them) == the matrix-by-matrix packer
            assert torch.equal(new["w8_feat_r6"], F.pack_gemm_r6(new["W8_feat"]))
            if with_render:
                for l in range(4):
                    assert torch.equal(new["R_r6"][l], F.pack_gemm_r6(new["R"][l][:, :spec.Kr if l == 0 else 256])), l
                for l in (1, 2, 3):
                    assert torch.equal(new["RT_r6"][l], F.pack_gemm_r6(new["RT"][l])), l
        # what the kernels require of the per-layer views: unit inner stride, 16-byte aligned rows
        for k in ("W", "WT") + (("R", "RT") if with_render else ()):
            for t in new[k]:
                if t is not None:
                    assert t.stride(-1) == 1 and t.stride(0) % 4 == 0 and t.data_ptr() % 16 == 0, k
    finally:
        hold_amd.set_precision(prev)


@pytest.mark.parametrize("kind", ["hand", "object"])
def test_pack_weights_f16x3_gemm_streams_match_the_matrix_by_matrix_packer(kind):
    """mode f16x3: the hold_gemm_h3 streams of the rendering net and of lin8's feature rows (one segmented maximum + one gather
    for all of them in pack_weights) == field.pack_gemm_h3 matrix by matrix, scales included; and pack_gemm_h3 is pack_gemm_r6's
    stream (rows, k order: pinned to the lane model in tests/test_r6_pack_cpu.py) in two fp16 limbs of the scaled matrix"""
    prev = hold_amd.precision()
    hold_amd.set_precision("f16x3")
    try:
        spec, iw, ib, rw, rb = _weights(kind, 5)
        pk = F.pack_weights(spec, iw, ib, rw, rb, True)
        for name in ("R_h3", "RT_h3", "c3_R", "c3_RT", "w8_feat_h3", "c3_w8", "R_r6", "RT_r6", "w8_feat_r6"):
            assert name in pk, name
        ref8, c8 = F.pack_gemm_h3(pk["W8_feat"])
        assert torch.equal(pk["w8_feat_h3"], ref8) and torch.equal(pk["c3_w8"], c8)
        for l in range(4):
            W = pk["R"][l][:, :spec.Kr if l == 0 else 256]
            refl, cl = F.pack_gemm_h3(W)
            assert torch.equal(pk["R_h3"][l], refl) and torch.equal(pk["c3_R"][l], cl), l
            WT = pk["RT"][l] if l else pk["R"][0][:, :256].t().contiguous()
            reft, ct = F.pack_gemm_h3(WT)
            assert torch.equal(pk["RT_h3"][l], reft) and torch.equal(pk["c3_RT"][l], ct), ("T", l)
            # two limbs of the scaled matrix, in pack_gemm_r6's order
            KS = (W.shape[1] + 63) // 64 * 4
            r6 = F.pack_gemm_r6(W).double().reshape(KS, 8, 3, 2, 32, 8).sum(2)
            h3 = refl.double().reshape(KS, 8, 2, 2, 32, 8)
            ws = r6 / float(cl)
            assert float(ws.abs().max()) >= 2.0 ** 13 and float(ws.abs().max()) < 2.0 ** 14
            assert torch.all((h3.sum(2) - ws).abs() <= ws.abs() * 2.0 ** -22 + 2.0 ** -25)
    finally:
        hold_amd.set_precision(prev)
