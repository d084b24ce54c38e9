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
ARITH_BWD = ARITH + ["r6"]  # descending sweeps: + hold_chain_r6, the register-resident structure (csrc/rchain.hip)


def _r6_stream(mode, mats):
    """weight stream of hold_chain_r6: DSP = field.pack_r6_stack of the 7 matrices, DBWD = field.pack_r6 (hold_trunk_r6's)"""
    from hold_amd import field as F
    if mode == "dsp":
        return F.pack_r6_stack(torch.stack(mats))
    return F.pack_r6(mats[0], torch.stack(mats[1:]))


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
            wpack_r6=_r6_stream("dsp", Ms) if arith == "r6" else None)
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


@pytest.mark.parametrize("P", [130, 128 * 257 + 3])
def test_chain_descending_dsp_background_skip_width(P):
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
    K.chain(K.CHAIN_DSP, P, v7, None, 7, 32, skip_layer=3, aux1=hs, out=out, wpack_r6=_r6_stream("dsp", Ms), skip_out=SKB)
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
    if arith == "r6":  # contract of hold_chain_r6 (DBWD): the skip layer's side columns come from aux2[3][:, 217:256]
        ts[3][:, SK:] = x0[:, :39]
    K.chain(K.CHAIN_DBWD, P, x0, _pack(Ws), 8, 5, skip_layer=3, side=x0, aux1=hs, aux2=ts, out=o1, out2=o2,
            wpack_x6=_pack_x6(Ws, 48) if arith == "f32x6" else None,
            wpack_r6=_r6_stream("dbwd", Ws) if arith == "r6" else None)
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
