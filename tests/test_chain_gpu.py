"""GPU parity of the LDS-resident layer chains (hold_chain) against a torch fp64 reference of the same sweeps, and
of the chain route through NodeField against the layer-by-layer GEMM route on the same weights."""
import pytest
import torch

pytestmark = pytest.mark.gpu

SK = 217


def _dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _pack(mats):
    """fragment order of include/hold_hip.h (hold_chain): [K/8][8 n-tiles][2 h][32 i][4]"""
    parts = []
    for m in mats:
        ch = m.shape[1] // 8
        parts.append(m.reshape(8, 32, ch, 2, 4).permute(2, 0, 3, 1, 4).reshape(-1))
    return torch.cat(parts).contiguous()


def _pack_x6(mats, first_k):
    from hold_amd import field as F
    return F.pack_x6(mats, first_k=first_k)


ARITH = ["f32", "f32x6"]  # hold_chain (fp32 MFMA) and hold_chain_x6 (3-limb bf16 split, fp32 accumulate): same tolerance
# descending / ascending sweeps: + hold_chain_r6, the register-resident structure (csrc/rchain.hip), and hold_chain_h3, the same
# structure in two fp16 limbs with per-point operand scales (csrc/rchain_h3.hip)
ARITH_BWD = ARITH + ["r6", "h3"]


def _r6_stream(mode, mats):
    """weight stream of hold_chain_r6: DSP = field.pack_r6_stack of the 7 matrices, DBWD = field.pack_r6 (hold_trunk_r6's)"""
    from hold_amd import field as F
    if mode == "dsp":
        return F.pack_r6_stack(torch.stack(mats))
    return F.pack_r6(mats[0], torch.stack(mats[1:]))


def _h3_stream(mode, mats):
    """operands of hold_chain_h3: (stream, c3 = 1 / s_w per chain layer); DSP = field.pack_h3_stack, DBWD = field.pack_h3"""
    from hold_amd import field as F
    if mode == "dsp":
        pk, sw = F.pack_h3_stack(torch.stack(mats))
    else:
        pk, sw = F.pack_h3(mats[0], torch.stack(mats[1:]))
    return dict(wpack_h3=pk, c3=(1.0 / sw).contiguous())


class _Guarded:
    """[P,256] output views followed by sentinel rows: rows >= P must never be written (hardware range check)."""

    def __init__(self, n, P, dev):
        self.big = [torch.full((P + 130, 256), 9.0, device=dev) for _ in range(n)]
        self.views = [b[:P] for b in self.big]
        self.P = P

    def check(self):
        for b in self.big:
            assert torch.all(b[self.P:] == 9.0)


def _sp(y):
    return torch.nn.functional.softplus(y, beta=100)


@pytest.mark.parametrize("arith", ARITH)
@pytest.mark.parametrize("P", [1, 130, 1000, 128 * 300 + 77])
def test_chain_softplus_forward(P, arith):
    from hold_amd import kernels as K
    dev = _dev()
    g = torch.Generator().manual_seed(P)
    x0 = torch.zeros(P, 40)
    x0[:, :39] = torch.randn(P, 39, generator=g)
    Ws = [torch.zeros(256, 40)] + [torch.randn(256, 256, generator=g) / 16 for _ in range(7)]
    Ws[0][:, :39] = torch.randn(256, 39, generator=g) / 6
    Ws[3][SK:] = 0  # layer 3 has 217 outputs
    bs = [torch.randn(256, generator=g) * 0.05 for _ in range(8)]
    x0, Ws, bs = x0.to(dev), [w.to(dev) for w in Ws], [b.to(dev) for b in bs]
    guard = _Guarded(8, P, dev)
    out = guard.views
    K.chain(K.CHAIN_SOFTPLUS, P, x0, _pack(Ws), 8, 5, skip_layer=3, side=x0, bias=bs, out=out,
            wpack_x6=_pack_x6(Ws, 48) if arith == "f32x6" else None)
    guard.check()
    cur = x0.double()
    for l in range(8):
        cur = _sp(cur @ Ws[l].double().t() + bs[l].double())
        if l == 3:
            cur = torch.cat([cur[:, :SK], x0[:, :39].double()], 1)
        err = (out[l].double() - cur).abs().max().item()
        assert err < 3e-5 * max(1.0, cur.abs().max().item()), (l, err)
        cur = out[l].double()  # follow the kernel's own rounding from layer to layer


@pytest.mark.parametrize("arith", ARITH_BWD)
@pytest.mark.parametrize("P,with_a2", [(130, False), (1000, True), (128 * 257 + 3, True), (128 * 257 + 3, False)])
def test_chain_descending_dsp(P, with_a2, arith):
    from hold_amd import kernels as K
    dev = _dev()
    g = torch.Generator().manual_seed(P + 5)
    v7 = torch.randn(P, 256, generator=g).to(dev)
    Ms = [(torch.randn(256, 256, generator=g) / 16).to(dev) for _ in range(7)]
    Ms[4][:, SK:] = 0  # the layer after the skip contracts over 217 inputs only
    hs = [_sp(torch.randn(P, 256, generator=g) * 0.03).to(dev) for _ in range(7)]
    a2 = [torch.randn(P, 256, generator=g).to(dev) for _ in range(7)] if with_a2 else None
    guard = _Guarded(7, P, dev)
    out = guard.views
    K.chain(K.CHAIN_DSP, P, v7, _pack(Ms), 7, 32, skip_layer=3, aux1=hs, aux2=a2, out=out,
            wpack_x6=_pack_x6(Ms, 256) if arith == "f32x6" else None,
            wpack_r6=_r6_stream("dsp", Ms) if arith in ("r6", "h3") else None, **(_h3_stream("dsp", Ms) if arith == "h3" else {}))
    guard.check()
    cur = v7.double()
    for j in range(7):
        y = cur @ Ms[j].double().t()
        r = y * (-torch.expm1(-100 * hs[j].double()))
        if with_a2:
            r = r + a2[j].double()
        if j == 3:
            r[:, SK:] = y[:, SK:]
        scale = max(1.0, r.abs().max().item())
        err = (out[j].double() - r).abs().max().item()
        assert err < 3e-5 * scale, (j, err)
        cur = out[j].double()


@pytest.mark.parametrize("arith", ["r6", "h3"])
@pytest.mark.parametrize("P", [130, 128 * 257 + 3])
def test_chain_descending_dsp_background_skip_width(P, arith):
    """hold_chain_r6 (DSP) with skip_out = 172, the background net's skip width (256 - 84 embedding columns,
    code/src/model/renderables/background.py): columns 172.. of the skip layer's output are the raw products"""
    from hold_amd import kernels as K
    dev = _dev()
    SKB = 172
    g = torch.Generator().manual_seed(P + 11)
    v7 = torch.randn(P, 256, generator=g).to(dev)
    Ms = [(torch.randn(256, 256, generator=g) / 16).to(dev) for _ in range(7)]
    Ms[4][:, SKB:] = 0  # the layer after the skip contracts over 172 inputs only
    hs = [_sp(torch.randn(P, 256, generator=g) * 0.03).to(dev) for _ in range(7)]
    guard = _Guarded(7, P, dev)
    out = guard.views
    K.chain(K.CHAIN_DSP, P, v7, None, 7, 32, skip_layer=3, aux1=hs, out=out, wpack_r6=_r6_stream("dsp", Ms), skip_out=SKB,
            **(_h3_stream("dsp", Ms) if arith == "h3" else {}))
    guard.check()
    cur = v7.double()
    for j in range(7):
        y = cur @ Ms[j].double().t()
        r = y * (-torch.expm1(-100 * hs[j].double()))
        if j == 3:
            r[:, SKB:] = y[:, SKB:]
        assert (out[j].double() - r).abs().max().item() < 3e-5 * max(1.0, r.abs().max().item()), j
        cur = out[j].double()


@pytest.mark.parametrize("arith", ARITH_BWD)
@pytest.mark.parametrize("P", [200, 128 * 256 + 64, 128 * 513 + 1])
def test_chain_second_order_dbwd(P, arith):
    from hold_amd import kernels as K
    dev = _dev()
    g = torch.Generator().manual_seed(P + 9)
    x0 = torch.zeros(P, 40)
    x0[:, :39] = torch.randn(P, 39, generator=g)
    Ws = [torch.zeros(256, 40)] + [torch.randn(256, 256, generator=g) / 16 for _ in range(7)]
    Ws[0][:, :39] = torch.randn(256, 39, generator=g) / 6
    Ws[3][SK:] = 0
    hs = [_sp(torch.randn(P, 256, generator=g) * 0.03).to(dev) for _ in range(8)]
    ts = [torch.randn(P, 256, generator=g).to(dev) for _ in range(8)]
    x0, Ws = x0.to(dev), [w.to(dev) for w in Ws]
    g1, g2 = _Guarded(8, P, dev), _Guarded(8, P, dev)
    o1, o2 = g1.views, g2.views
    if arith in ("r6", "h3"):  # contract of hold_chain_r6 / _h3 (DBWD): the skip layer's side columns come from aux2[3][:, 217:256]
        ts[3][:, SK:] = x0[:, :39]
    K.chain(K.CHAIN_DBWD, P, x0, _pack(Ws), 8, 5, skip_layer=3, side=x0, aux1=hs, aux2=ts, out=o1, out2=o2,
            wpack_x6=_pack_x6(Ws, 48) if arith == "f32x6" else None,
            wpack_r6=_r6_stream("dbwd", Ws) if arith in ("r6", "h3") else None, **(_h3_stream("dbwd", Ws) if arith == "h3" else {}))
    g1.check()
    g2.check()
    cur = x0.double()
    for l in range(8):
        y = cur @ Ws[l].double().t()
        e = torch.exp(-100 * hs[l].double())
        r = y * (1 - e)
        r2 = 100 * y * ts[l].double() * e
        if l == 3:
            r = torch.cat([r[:, :SK], x0[:, :39].double()], 1)
            r2[:, SK:] = 0
        s1 = max(1.0, r.abs().max().item())
        s2 = max(1.0, r2.abs().max().item())
        assert (o1[l].double() - r).abs().max().item() < 3e-5 * s1, l
        assert (o2[l].double() - r2).abs().max().item() < 3e-5 * s2, l
        cur = o1[l].double()


def test_chain_r6_optional_outputs():
    """hold_chain_r6 DSP with some out[] entries NULL (the eval-mode reverse sweep keeps t_0 and t_3 only): the stored
    layers equal the all-outputs run bit for bit"""
    from hold_amd import kernels as K
    dev = _dev()
    P = 777
    g = torch.Generator().manual_seed(21)
    v7 = torch.randn(P, 256, generator=g).to(dev)
    Ms = [(torch.randn(256, 256, generator=g) / 16).to(dev) for _ in range(7)]
    hs = [_sp(torch.randn(P, 256, generator=g) * 0.03).to(dev) for _ in range(7)]
    full = [torch.empty(P, 256, device=dev) for _ in range(7)]
    r6 = _r6_stream("dsp", Ms)
    K.chain(K.CHAIN_DSP, P, v7, _pack(Ms), 7, 32, skip_layer=3, aux1=hs, out=full, wpack_r6=r6)
    part = [torch.full((P, 256), 5.0, device=dev) if j in (3, 6) else None for j in range(7)]
    K.chain(K.CHAIN_DSP, P, v7, _pack(Ms), 7, 32, skip_layer=3, aux1=hs, out=part, wpack_r6=r6)
    for j in (3, 6):
        assert torch.equal(part[j], full[j])


def test_chain_row_split_matches_single_call(monkeypatch):
    """batches beyond the kernel's 32-bit offset range are split by rows in hold_amd/kernels.py:chain"""
    from hold_amd import kernels as K
    dev = _dev()
    P = 1000
    g = torch.Generator().manual_seed(11)
    v7 = torch.randn(P, 256, generator=g).to(dev)
    Ms = [(torch.randn(256, 256, generator=g) / 16).to(dev) for _ in range(7)]
    hs = [_sp(torch.randn(P, 256, generator=g) * 0.03).to(dev) for _ in range(7)]
    ref = [torch.empty(P, 256, device=dev) for _ in range(7)]
    K.chain(K.CHAIN_DSP, P, v7, _pack(Ms), 7, 32, skip_layer=3, aux1=hs, out=ref)
    monkeypatch.setattr(K, "_CHAIN_MAX_ROWS", 384)
    out = [torch.empty(P, 256, device=dev) for _ in range(7)]
    K.chain(K.CHAIN_DSP, P, v7, _pack(Ms), 7, 32, skip_layer=3, aux1=hs, out=out)
    for a, b in zip(ref, out):
        assert torch.equal(a, b)


def test_chain_rejects_bad_arguments():
    import ctypes as C
    from hold_amd import _lib
    L = _lib.lib()
    d = _lib.ChainDesc()
    assert L.hold_chain(C.byref(d), None) == -1
    assert L.hold_chain(None, None) == -1
    assert L.hold_chain_pack_floats(5, 8) == (5 + 7 * 32) * 2048
    assert L.hold_chain_pack_floats(7, 8) == -1


@pytest.mark.parametrize("kind,node", [("hand", "right"), ("object", "object")])
def test_field_chain_route_matches_layered_route(kind, node):
    """NodeField forward + backward (all parameter / pose gradients) with hold_chain vs one hold_gemm_nt per layer."""
    from hold_amd import field as F, synthetic as syn
    dev = _dev()
    sc = syn.make_scene(2)
    sd = {k: torch.as_tensor(v).to(dev) for k, v in syn.make_state_dict(sc, perturb=0.05).items()}
    spec = F.FieldSpec(kind)
    pre = f"nodes.{node}."
    eff = lambda p: sd[p + ".weight_v"] * (sd[p + ".weight_g"] / sd[p + ".weight_v"].norm(dim=1, keepdim=True))
    iw = [eff(pre + f"implicit_network.lin{l}") for l in range(9)]
    ib = [sd[pre + f"implicit_network.lin{l}.bias"] for l in range(9)]
    rw = [eff(pre + f"rendering_network.lin{l}") for l in range(5)]
    rb = [sd[pre + f"rendering_network.lin{l}.bias"] for l in range(5)]
    pk = F.pack_weights(spec, iw, ib, rw, rb, need_bwd=True)
    B, ppf = 2, 333
    P = B * ppf
    g = torch.Generator().manual_seed(3)
    x = torch.zeros(P, 4, device=dev)
    # points in a shell around the geometric-init sphere: at the centre |grad sdf| -> 0 and the normalised gradient (and
    # everything downstream of it) amplifies rounding differences between the two routes without bound
    dirs = torch.nn.functional.normalize(torch.randn(P, 3, generator=g), dim=1)
    x[:, :3] = (dirs * (0.25 + 0.35 * torch.rand(P, 1, generator=g))).to(dev)
    nb = spec.n_bones
    tfs = torch.eye(4).reshape(1, 1, 16).repeat(B, nb, 1)
    tfs[:, :, [3, 7, 11]] += torch.randn(B, nb, 3, generator=g) * 0.01
    dfm = dict(tfs=tfs.to(dev).contiguous())
    if kind == "hand":
        verts = (torch.nn.functional.normalize(torch.randn(B, 778, 3, generator=g), dim=-1) * 0.4).to(dev).contiguous()
        skin = torch.rand(778, 16, generator=g)
        dfm.update(verts=verts, verts_c=verts[:1].contiguous(), skin_w=(skin / skin.sum(1, keepdim=True)).to(dev).contiguous())
    barf = torch.rand(39, generator=g).to(dev) if kind == "object" else None
    pose = torch.randn(B, 8, generator=g).to(dev)
    tcode = torch.randn(B, 32, generator=g).to(dev) if kind == "object" else None
    d_sdf = torch.randn(P, generator=g).to(dev)
    d_rgb = torch.randn(P, 3, generator=g).to(dev)
    d_n = torch.randn(P, 3, generator=g).to(dev)
    res = {}
    for route in (False, True):
        F.USE_CHAIN = route
        nf = F.NodeField(spec, dev)
        o = nf.forward(pk, x, P, ppf, dfm, barf, pose, tcode, training=True)
        fw = {k: o[k].clone() for k in ("sdf", "rgb", "normal", "grad")}
        gr = nf.backward(d_sdf, d_rgb, d_n, B)
        res[route] = (fw, gr, [(r > 0).clone() for r in nf.saved["r"]])
    F.USE_CHAIN = True
    for k in res[False][0]:
        a, b = res[False][0][k], res[True][0][k]
        assert (a - b).abs().max().item() <= 2e-5 * max(1.0, a.abs().max().item()), k
    ga, gb = res[False][1], res[True][1]
    # The two routes differ by rounding only in the trunk (chain: split-precision or fp32 MFMA in another summation
    # order).  Downstream of it sits the rendering net's ReLU: a pre-activation within rounding of zero switches its
    # unit on in one route and off in the other, which moves that unit's share of a weight gradient discontinuously
    # -- a few such flips among 666 points x 1 024 units bound the agreement of the rendering-net gradients at ~1 %.
    # The comparison is strict when the two routes took the same side of every ReLU; the (point, unit) pairs on which they
    # did not are counted from the stored activations, must be a handful, and only then are the bounds relaxed: a flipped
    # pair moves one point's contribution to one row of its layer's weight gradient and a rank-one term of everything
    # upstream of it.
    flips = sum(int((m0 != m1).sum()) for m0, m1 in zip(res[False][2], res[True][2]))
    assert flips <= 12, flips
    t_max = 5e-4 if flips == 0 else 5e-2
    t_nrm = 2e-4 if flips == 0 else 1e-2
    for k in ("iw", "ib"):
        for i, (a, b) in enumerate(zip(ga[k], gb[k])):
            assert (a - b).abs().max().item() <= (2e-4 if flips == 0 else 5e-3) * max(1e-3, a.abs().max().item()), (k, i, flips)
    for k in ("rw", "rb"):
        for i, (a, b) in enumerate(zip(ga[k], gb[k])):
            scale = max(1e-3, a.abs().max().item())
            d = (a - b).abs().reshape(-1)
            assert (d > 5e-4 * scale).float().mean().item() <= (0.0 if flips == 0 else 0.004 * flips + 0.02), (k, i, flips)
            assert d.max().item() <= t_max * scale, (k, i, flips)
            assert (a - b).norm().item() <= t_nrm * max(1e-3, a.norm().item()), (k, i, flips)
    for k in ("tfs", "pose_embed", "time_code"):
        if ga[k] is None:
            continue
        # pose_embed / time_code enter the rendering net's first layer: their gradients sit behind the same ReLU flips
        t = (2e-4 if k == "tfs" else 2e-3) * (1 if flips == 0 else 10)
        assert (ga[k] - gb[k]).abs().max().item() <= t * max(1e-3, ga[k].abs().max().item()), (k, flips)


def _sweep_case(P, seed, dev, row_scale=None, a2_scale=1.0):
    """operands of the three sweeps on P points: weights like the trunk's (N(0, 1/16)), softplus outputs, N(0,1) cotangents
    optionally scaled row by row (a point's cotangent has ITS magnitude: compositing weights span many orders of magnitude)"""
    g = torch.Generator().manual_seed(seed)
    rs = torch.ones(P, 1) if row_scale is None else row_scale.view(P, 1)
    Ms = [(torch.randn(256, 256, generator=g) / 16).to(dev) for _ in range(7)]
    Ms[4][:, SK:] = 0
    v7 = (torch.randn(P, 256, generator=g) * rs).to(dev)
    hs = [_sp(torch.randn(P, 256, generator=g) * 0.03).to(dev) for _ in range(8)]
    a2 = [(torch.randn(P, 256, generator=g) * rs * a2_scale).to(dev) for _ in range(7)]
    x0 = torch.zeros(P, 40)
    x0[:, :39] = torch.randn(P, 39, generator=g) * rs
    Ws = [torch.zeros(256, 40)] + [torch.randn(256, 256, generator=g) / 16 for _ in range(7)]
    Ws[0][:, :39] = torch.randn(256, 39, generator=g) / 6
    Ws[3][SK:] = 0
    ts = [(torch.randn(P, 256, generator=g)).to(dev) for _ in range(8)]
    x0, Ws = x0.to(dev), [w.to(dev) for w in Ws]
    ts[3][:, SK:] = x0[:, :39]
    return Ms, v7, hs, a2, x0, Ws, ts


def _run_sweeps(P, case, arith, dev):
    """-> (DSP outs, DSP + a2 outs, DBWD out, DBWD out2) of one arithmetic ('r6' / 'h3')"""
    from hold_amd import kernels as K
    Ms, v7, hs, a2, x0, Ws, ts = case
    kw_d = dict(wpack_r6=_r6_stream("dsp", Ms), **(_h3_stream("dsp", Ms) if arith == "h3" else {}))
    kw_b = dict(wpack_r6=_r6_stream("dbwd", Ws), **(_h3_stream("dbwd", Ws) if arith == "h3" else {}))
    new = lambda n: [torch.empty(P, 256, device=dev) for _ in range(n)]
    o_t, o_r, o_v, o_a = new(7), new(7), new(8), new(8)
    K.chain(K.CHAIN_DSP, P, v7, None, 7, 32, skip_layer=3, aux1=hs[:7], out=o_t, **kw_d)
    K.chain(K.CHAIN_DSP, P, v7, None, 7, 32, skip_layer=3, aux1=hs[:7], aux2=a2, out=o_r, **kw_d)
    K.chain(K.CHAIN_DBWD, P, x0, None, 8, 5, skip_layer=3, side=x0, aux1=hs, aux2=ts, out=o_v, out2=o_a, **kw_b)
    return o_t, o_r, o_v, o_a


def _ref_sweeps(case, sl):
    """fp64 restatement of the three sweeps on the rows `sl` (teacher-forced: every layer from the fp64 result of the one before)"""
    Ms, v7, hs, a2, x0, Ws, ts = case
    t, r = [], []
    for with_a2, dst in ((False, t), (True, r)):
        cur = v7[sl].double()
        for j in range(7):
            y = cur @ Ms[j].double().t()
            o = y * (-torch.expm1(-100 * hs[j][sl].double()))
            if with_a2:
                o = o + a2[j][sl].double()
            if j == 3:
                o[:, SK:] = y[:, SK:]
            dst.append(o)
            cur = o
    v, aa = [], []
    cur = x0[sl].double()
    for l in range(8):
        y = cur @ Ws[l].double().t()
        e = torch.exp(-100 * hs[l][sl].double())
        o, o2 = y * (1 - e), 100 * y * ts[l][sl].double() * e
        if l == 3:
            o = torch.cat([o[:, :SK], x0[sl, :39].double()], 1)
            o2[:, SK:] = 0
        v.append(o)
        aa.append(o2)
        cur = o
    return t, r, v, aa


def test_chain_h3_error_against_fp64_is_within_one_and_a_half_of_r6_on_half_a_million_points():
    """the gate of the f16x3 sweeps: on 524 288 points the error of t (descending sweep of the normal path),
    r (first-order backward with the additive side input) and vbar / a2 (second-order ascending sweep) against fp64 is at most
    1.5 x that of the exact three-limb bf16 kernels (hold_chain_r6), max and rms, END TO END through the chain (each kernel feeds
    its own layers) -- (a) with O(1) cotangents and (b) with per-point magnitudes spread log-uniformly over 1e-12 .. 1e+2
    (a point's cotangent has its own magnitude: compositing weights along a ray), where the error of a point is measured
    against THAT POINT's largest value of the layer (a common absolute bound would only see the largest points)."""
    from hold_amd import kernels as K
    dev = _dev()
    P = 524288
    g = torch.Generator().manual_seed(77)
    for name, rs in (("unit", None), ("spread", 10.0 ** (torch.rand(P, generator=g) * 14 - 12))):
        case = _sweep_case(P, 31, dev, rs)
        n0 = K.h3_overflow_count(dev)
        o6, o3 = _run_sweeps(P, case, "r6", dev), _run_sweeps(P, case, "h3", dev)
        assert K.h3_overflow_count(dev) == n0, "the f16x3 sweeps fell back to f32x6 on well-behaved operands"
        worst = {}
        for c0 in range(0, P, 65536):
            sl = slice(c0, c0 + 65536)
            ref = _ref_sweeps(case, sl)
            for fam, (a6, a3, rf) in enumerate(zip(o6, o3, ref)):
                for l, (x6, x3, xr) in enumerate(zip(a6, a3, rf)):
                    den = xr.abs().amax(1, keepdim=True).clamp_min(1e-300)  # the point's own scale
                    e6, e3 = ((x6[sl].double() - xr).abs() / den), ((x3[sl].double() - xr).abs() / den)
                    w = worst.setdefault(fam, [0.0, 0.0, 0.0, 0.0, 0])
                    w[0], w[1] = max(w[0], float(e6.max())), max(w[1], float(e3.max()))
                    w[2], w[3], w[4] = w[2] + float(e6.pow(2).sum()), w[3] + float(e3.pow(2).sum()), w[4] + e6.numel()
        for fam, w in worst.items():
            r6m, h3m, r6r, h3r = w[0], w[1], (w[2] / w[4]) ** 0.5, (w[3] / w[4]) ** 0.5
            print(f"{name} family {('t', 'r', 'vbar', 'a2')[fam]}: r6 max {r6m:.3e} rms {r6r:.3e} | h3 max {h3m:.3e} rms {h3r:.3e}")
            assert h3m <= 1.5 * r6m and h3r <= 1.5 * r6r, (name, fam, w)


def test_chain_h3_overflow_falls_back_to_r6_on_the_device():
    """a point whose running value grows by more than the 2^9 headroom of its predicted scale within ONE layer (here: an
    additive side input 1e5 x the running cotangent at layer 2 on a few rows) leaves fp16's
    range: the launch sets the guard, the conditional hold_chain_r6 launch behind it recomputes it, the results are hold_chain_r6's
    bit for bit and the event is counted; a well-behaved launch afterwards is the f16x3 kernel's again."""
    from hold_amd import kernels as K
    dev = _dev()
    P = 128 * 257 + 3
    case = _sweep_case(P, 5, dev)
    Ms, v7, hs, a2, x0, Ws, ts = case
    a2[2][1000:1004] *= 1e5
    guard = K.h3_guard(dev)
    n0 = K.h3_overflow_count(dev)
    o6, o3 = _run_sweeps(P, case, "r6", dev), _run_sweeps(P, case, "h3", dev)
    assert K.h3_overflow_count(dev) == n0 + 1 and guard[:2].tolist() == [0, 0]  # only the sweep that reads a2 overflowed
    for x6, x3 in zip(o6[1], o3[1]):
        assert torch.equal(x6, x3)
    assert not all(torch.equal(a, b) for a, b in zip(o6[0], o3[0]))  # the others ran in f16x3
    a2[2][1000:1004] *= 1e-5
    o3b = _run_sweeps(P, case, "h3", dev)
    assert K.h3_overflow_count(dev) == n0 + 1
    ref = _ref_sweeps(case, slice(0, 4096))
    for fam in range(4):
        for x3, xr in zip(o3b[fam], ref[fam]):
            assert (x3[:4096].double() - xr).abs().max().item() < 3e-5 * max(1.0, xr.abs().max().item())
