"""GPU parity of the MFMA GEMM kernels against a torch fp64 reference of the same op."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


@pytest.fixture(params=["f16x3", "f32x6", "f32"])
def arith(request):
    """run a test under both arithmetics: hold_gemm_nt_x6 / hold_gemm_nt (same tolerances)"""
    import hold_amd
    prev = hold_amd.precision()
    hold_amd.set_precision(request.param)
    yield request.param
    hold_amd.set_precision(prev)


@pytest.fixture(params=["f16x3", "f32x6"])
def split_arith(request):
    """the two split-precision modes: in f16x3 the whole-dW weight gradients run the two-limb fp16 kernel (hold_wgrad_h3 /
    hold_wgrad_group_h3), everything else the three-limb bf16 kernels of f32x6 -- same tests, same tolerances"""
    import hold_amd
    prev = hold_amd.precision()
    hold_amd.set_precision(request.param)
    yield request.param
    hold_amd.set_precision(prev)


def _softplus_ref(y):
    return torch.nn.functional.softplus(y, beta=100)


@pytest.mark.parametrize("P,N,K", [(1000, 256, 256), (777, 217, 256), (130, 257, 256), (4096, 3, 256), (513, 256, 40),
                                    (300, 39, 256), (260, 256, 272), (64, 128, 316)])
def test_gemm_nt_plain_and_softplus(P, N, K, arith):
    from hold_amd import gemm
    dev = _dev()
    g = torch.Generator(device="cpu").manual_seed(P + N + K)
    A = torch.randn(P, K, generator=g).to(dev)
    W = (torch.randn(N, K, generator=g) / K ** 0.5).to(dev)
    b = torch.randn(N, generator=g).to(dev) * 0.1
    ref = A.double() @ W.double().t() + b.double()
    out = torch.full((P, N + 5), -7.0, device=dev)
    gemm.gemm_nt(A, W, out[:, :N], bias=b, N=N)
    assert torch.all(out[:, N:] == -7.0)
    err = (out[:, :N].double() - ref).abs().max().item()
    assert err < 2e-5, err
    # asymmetric check on layout: columns and rows must not be swapped
    out2 = torch.empty(P, N, device=dev)
    gemm.gemm_nt(A, W, out2, bias=b, epi=gemm.EPI_SOFTPLUS)
    ref2 = _softplus_ref(ref * 0.05) if False else _softplus_ref(ref)
    assert (out2.double() - ref2).abs().max().item() < 2e-5


def test_gemm_split_and_epilogues(arith):
    from hold_amd import gemm
    dev = _dev()
    torch.manual_seed(0)
    P, N, K = 515, 256, 256
    A = torch.randn(P, K, device=dev)
    W = torch.randn(N, K, device=dev) / 16
    H = torch.nn.functional.softplus(torch.randn(P, 217, device=dev) * 0.05, beta=100)
    add = torch.randn(P, 217, device=dev)
    y = (A.double() @ W.double().t()) * 0.7071
    out = torch.empty(P, 217, device=dev)
    raw = torch.empty(P, 39, device=dev)
    gemm.gemm_nt(A, W, out, epi=gemm.EPI_MUL_DSP, alpha=0.7071, n_split=217, out_raw=raw, aux1=H, aux2=add)
    s = -torch.expm1(-100 * H.double())
    assert (out.double() - (y[:, :217] * s + add.double())).abs().max().item() < 3e-5
    assert (raw.double() - y[:, 217:]).abs().max().item() < 3e-5
    # DBWD
    Hf = torch.nn.functional.softplus(torch.randn(P, N, device=dev) * 0.05, beta=100)
    T = torch.randn(P, N, device=dev)
    o1 = torch.empty(P, N, device=dev)
    o2 = torch.empty(P, N, device=dev)
    gemm.gemm_nt(A, W, o1, epi=gemm.EPI_DBWD, aux1=Hf, aux2=T, out2=o2)
    y = A.double() @ W.double().t()
    s = -torch.expm1(-100 * Hf.double())
    assert (o1.double() - y * s).abs().max().item() < 3e-5
    ref2 = 100 * y * T.double() * (1 - s)
    assert ((o2.double() - ref2).abs().max() / ref2.abs().max()).item() < 1e-5
    # relu mask / sigmoid / accumulate
    o3 = torch.empty(P, N, device=dev)
    gemm.gemm_nt(A, W, o3, epi=gemm.EPI_MUL_DRELU, aux1=T)
    assert (o3.double() - y * (T > 0)).abs().max().item() < 3e-5
    o4 = torch.ones(P, N, device=dev)
    gemm.gemm_nt(A, W, o4, accumulate=True)
    assert (o4.double() - (y + 1)).abs().max().item() < 3e-5
    o5 = torch.empty(P, N, device=dev)
    gemm.gemm_nt(A, W, o5, epi=gemm.EPI_SIGMOID)
    assert (o5.double() - torch.sigmoid(y)).abs().max().item() < 1e-5


@pytest.mark.parametrize("P,N,K", [(5000, 256, 256), (1234, 217, 256), (999, 257, 256), (4100, 3, 256), (700, 256, 40),
                                    (70, 256, 304)])
def test_wgrad(P, N, K):
    from hold_amd import gemm
    dev = _dev()
    torch.manual_seed(P)
    R = torch.randn(P, N, device=dev)
    X = torch.randn(P, K, device=dev)
    dW = torch.ones(N, K, device=dev)
    db = torch.ones(N, device=dev)
    gemm.wgrad(R, X, dW, db, accumulate=True)
    ref = R.double().t() @ X.double() + 1
    refb = R.double().sum(0) + 1
    assert ((dW.double() - ref).abs().max() / ref.abs().max()).item() < 1e-5
    assert ((db.double() - refb).abs().max() / refb.abs().max()).item() < 1e-5


@pytest.mark.parametrize("P,ldr,ldx,bias", [(4096, 256, 256, True), (4144, 256, 256, False), (200000, 256, 256, True),
                                            (65536 + 16, 304, 272, True)])
def test_wgrad_256x256_whole_layer_workgroups(P, ldr, ldx, bias, split_arith):
    """the register-resident 256 x 256 weight gradient (csrc/wgrad_r6.hip: taken by hold_wgrad_x6 for N = K = 256 and P a
    multiple of 16 >= 4096) against fp64 -- row strides wider than the matrix, with / without the bias sums, accumulate"""
    import hold_amd
    from hold_amd import gemm
    dev = _dev()
    torch.manual_seed(P)
    Rb = torch.randn(P, ldr, device=dev)
    Xb = torch.randn(P, ldx, device=dev)
    R, X = Rb[:, :256], Xb[:, :256]
    dW = torch.ones(256, 256, device=dev)
    db = torch.ones(256, device=dev) if bias else None
    gemm.wgrad(R, X, dW, db, accumulate=True)
    ref = R.double().t() @ X.double() + 1
    assert ((dW.double() - ref).abs().max() / ref.abs().max()).item() < 1e-5
    if bias:
        refb = R.double().sum(0) + 1
        assert ((db.double() - refb).abs().max() / refb.abs().max()).item() < 1e-5
    # structured operands: a wrong row / column / limb pairing cannot hide behind random cancellation
    R.zero_(); X.zero_()
    R[:, 37] = 1.0
    X[:, 201] = torch.arange(P, device=dev, dtype=torch.float32) % 7 + 0.123456789
    dW2 = torch.empty(256, 256, device=dev)
    gemm.wgrad(R, X, dW2, None)
    ref2 = torch.zeros(256, 256, dtype=torch.float64)
    ref2[37, 201] = float(X[:, 201].double().sum())
    # 2e-6: these are sums of up to 200 000 POSITIVE products per entry -- the fp32 accumulation error of such a sum has no
    # cancellation to hide behind: measured on the hardware (round 5, GPU call 4) 1.07e-6 for the true-fp32 MFMA kernel at
    # P = 65 552 and 1.01e-6 for the two-limb fp16 kernel at P = 200 000 (2.7e-9 at P = 65 552), 1.9e-7 .. 1e-6 for the three-
    # limb bf16 kernel; the representation error of these operands is 5e-9 in every arithmetic
    assert float((dW2.double().cpu() - ref2).abs().max()) < 2e-6 * ref2[37, 201]


@pytest.mark.parametrize("P,N,K,ldr,ldx", [(4096 * 3 + 16, 217, 256, 256, 256), (65536, 217, 256, 256, 256), (8192, 256, 272, 256, 272),
                                           (20000 * 16, 256, 304, 256, 304), (4096, 130, 256, 260, 256), (4096, 256, 320, 256, 320)])
def test_wgrad_whole_layer_workgroups_other_shapes(P, N, K, ldr, ldx, split_arith):
    """the whole-dW weight gradient for the trunk's 217-row layer (N < 256: the kernel reads 256 columns of R and the rows
    >= N of its partial tiles are never reduced) and the rendering net's first layer (K = 272 / 304: 256 columns this way, the
    tail by the tile kernel) against fp64 -- with the columns N..255 of R POISONED (NaN / Inf: round 3's 0 x NaN in the
    branch-free bias sums), sentinels around dW / db, accumulate on and off"""
    import hold_amd
    from hold_amd import gemm
    dev = _dev()
    torch.manual_seed(P + N + K)
    Rb = torch.randn(P, ldr, device=dev)
    Xb = torch.randn(P, ldx, device=dev)
    if N < 256:
        Rb[:, N:256] = float("nan")
        Rb[::3, N:256] = float("inf")
    R, X = Rb[:, :N], Xb[:, :K]
    big = torch.full((N + 2, K + 4), 7.0, device=dev)  # dW as an interior view: rows / columns beyond stay untouched
    dW = big[1:N + 1, :K]
    dbb = torch.full((N + 8,), 7.0, device=dev)
    db = dbb[4:4 + N]
    dW.fill_(1.0)
    db.fill_(1.0)
    gemm.wgrad(R, X, dW, db, accumulate=True)
    ref = R.double().t() @ X.double() + 1
    refb = R.double().sum(0) + 1
    assert torch.isfinite(dW).all() and torch.isfinite(db).all()
    assert ((dW.double() - ref).abs().max() / ref.abs().max()).item() < 1e-5
    assert ((db.double() - refb).abs().max() / refb.abs().max()).item() < 1e-5
    assert torch.all(big[0] == 7.0) and torch.all(big[N + 1] == 7.0) and torch.all(big[:, K:] == 7.0)
    assert torch.all(dbb[:4] == 7.0) and torch.all(dbb[4 + N:] == 7.0)
    dW2 = torch.empty(N, K, device=dev)
    gemm.wgrad(R, X, dW2, None)
    assert ((dW2.double() - (ref - 1)).abs().max() / ref.abs().max()).item() < 1e-5


@pytest.mark.gpu
@pytest.mark.parametrize("P", [4096, 20000, 131072 + 48])
def test_wgrad_group_matches_fp64_and_single_launches(P, split_arith):
    """hold_wgrad_group_x6: several (R, X) pairs over the same points in one launch -- two pairs sharing a destination (one
    with a bias, one without; the 217-row layer with its columns 217..255 poisoned), single pairs with and without bias,
    accumulate on and off -- against fp64 (the grouped kernel splits the points differently from hold_wgrad_x6, so the
    comparison with a single launch is at 1e-5 too, not bit for bit), with sentinels around every destination."""
    import hold_amd
    from hold_amd import gemm
    dev = _dev()
    torch.manual_seed(P)
    mk = lambda ld=256: torch.randn(P, ld, device=dev)
    R = [mk(), mk(), mk(), mk(260), mk(), mk()]
    X = [mk(), mk(), mk(), mk(), mk(272), mk()]
    R[2][:, 217:] = float("nan")
    R[3][:, 217:256] = float("inf")
    big = torch.full((5, 258, 260), 7.0, device=dev)
    dW = [big[i, 1:257, :256] for i in range(4)]
    dW[1] = big[1, 1:218, :256]
    for d in dW:
        d.fill_(1.0)
    db = [torch.full((256,), 1.0, device=dev) for _ in range(4)]
    grp = gemm.WgradGroup()
    grp.add(R[0], X[0], dW[0], db[0], accumulate=True)            # destination 0: two pairs, bias from the first only
    grp.add(R[2], X[2], dW[1], None, N=217, accumulate=True)      # destination 1 (217 rows): two pairs, bias from the second
    grp.add(R[1], X[1], dW[0], None, accumulate=True)
    grp.add(R[3], X[3], dW[1], db[1][:217], N=217, accumulate=True)
    grp.add(R[4], X[4][:, :256], dW[2], db[2])                    # single pair, overwrite, X wider in memory
    grp.add(R[5], X[5], dW[3], None, accumulate=True)             # single pair, no bias
    grp.flush()
    dd = lambda a, b: a.double().t() @ b.double()
    ref = [dd(R[0], X[0]) + dd(R[1], X[1]) + 1, dd(R[2][:, :217], X[2]) + dd(R[3][:, :217], X[3]) + 1,
           dd(R[4], X[4][:, :256]), dd(R[5], X[5]) + 1]
    refb = [R[0].double().sum(0) + 1, R[3][:, :217].double().sum(0) + 1, R[4].double().sum(0)]
    for i in range(4):
        assert torch.isfinite(dW[i]).all(), i
        assert ((dW[i].double() - ref[i]).abs().max() / ref[i].abs().max()).item() < 1e-5, i
    assert ((db[0].double() - refb[0]).abs().max() / refb[0].abs().max()).item() < 1e-5
    assert ((db[1][:217].double() - refb[1]).abs().max() / refb[1].abs().max()).item() < 1e-5
    assert torch.all(db[1][217:] == 1.0)
    assert ((db[2].double() - refb[2]).abs().max() / refb[2].abs().max()).item() < 1e-5
    assert torch.all(db[3] == 1.0)
    # nothing outside the destinations was touched
    assert torch.all(big[4] == 7.0) and torch.all(big[:, 0] == 7.0) and torch.all(big[:, 257] == 7.0)
    assert torch.all(big[:, :, 256:] == 7.0) and torch.all(big[1, 218:257] == 7.0)
    # the same pairs one launch each
    one = torch.zeros(256, 256, device=dev)
    gemm.wgrad(R[5], X[5], one, None)
    assert ((one.double() + 1 - dW[3].double()).abs().max() / ref[3].abs().max()).item() < 1e-5
    # run-to-run determinism of the grouped reduction
    again = [torch.ones(256, 256, device=dev), torch.ones(256, device=dev)]
    g2 = gemm.WgradGroup()
    g2.add(R[0], X[0], again[0], again[1], accumulate=True)
    g2.add(R[1], X[1], again[0], None, accumulate=True)
    g2.add(R[5], X[5], torch.empty(256, 256, device=dev), None)
    g2.flush()
    assert ((again[0].double() - ref[0]).abs().max() / ref[0].abs().max()).item() < 1e-5
    assert ((again[1].double() - refb[0]).abs().max() / refb[0].abs().max()).item() < 1e-5


@pytest.mark.parametrize("P,N,where,acc", [(4096 + 17, 39, "t3", True), (8192, 39, "t3", False), (5000, 40, "own", True),
                                          (4099, 16, "rin", False), (131072 + 5, 48, "rin", False), (4128, 64, "own", True),
                                          (4096, 1, "own", False), (70001, 33, "own", True)])
def test_narrow_gemm_matches_fp64_and_the_tile_kernel(P, N, where, acc, split_arith):
    """hold_gemm_narrow_x6 (csrc/rnarrow.hip): C[:, :N] (+)= A[:, :256] W[:N, :256]^T for the shapes of the path -- N = 39 into
    the 4-byte-aligned columns 217.. of a 256-wide buffer (d sdf / d embedding in place in t_3), N = 16 / 48 into the columns
    256.. of the colour net's input gradient, N = 40 into a buffer of its own, the edge widths 1 / 33 / 64 -- with a partial last
    32-point tile, accumulate on and off, sentinels around the destination and W rows beyond N poisoned; against fp64 and
    against hold_gemm_nt_x6"""
    import hold_amd
    from hold_amd import gemm
    dev = _dev()
    torch.manual_seed(P + N)
    A = torch.randn(P, 260, device=dev)[:, :256] if N % 2 else torch.randn(P, 256, device=dev)
    Wb = torch.randn(72, 256, device=dev) / 16
    Wb[N:] = float("nan")
    W = Wb[:N]
    if where == "t3":
        host = torch.full((P + 1, 256), 7.0, device=dev)
        C = host[:P, 217:217 + N]
    elif where == "rin":
        host = torch.full((P + 1, 256 + N + 4), 7.0, device=dev)
        C = host[:P, 256:256 + N]
    else:
        host = torch.full((P + 1, N + 3), 7.0, device=dev)
        C = host[:P, :N]
    C.copy_(torch.randn(P, N, device=dev))
    C0 = C.clone()
    ref = A.double() @ W.double().t() + (C0.double() if acc else 0)
    C_nt = C0.clone()
    gemm.gemm_nt(A, W, C_nt, N=N, K=256, accumulate=acc)
    gemm.gemm_narrow(A, W, C, N=N, accumulate=acc)
    torch.cuda.synchronize()
    sc = float(ref.abs().max())
    assert torch.isfinite(C).all()
    assert float((C.double() - ref).abs().max()) < 2e-6 * sc, float((C.double() - ref).abs().max()) / sc
    assert float((C - C_nt).abs().max()) < 2e-6 * sc
    # nothing outside the N columns of the P rows was touched
    keep = torch.ones_like(host, dtype=torch.bool)
    c0 = 217 if where == "t3" else (256 if where == "rin" else 0)
    keep[:P, c0:c0 + N] = False
    assert torch.all(host[keep] == 7.0)


def test_narrow_gemm_rejects_what_it_cannot_take():
    import ctypes as C
    from hold_amd import _lib
    L = _lib.lib()
    p = C.c_void_p(256)
    assert L.hold_gemm_narrow_x6(p, 256, p, 256, p, 64, 4096, 65, 0, None) != 0   # wider than 64
    assert L.hold_gemm_narrow_x6(p, 128, p, 256, p, 64, 4096, 39, 0, None) != 0   # A narrower than K = 256 in memory
    assert L.hold_gemm_narrow_x6(p, 256, p, 256, p, 16, 4096, 39, 0, None) != 0   # ldc < N
    assert L.hold_gemm_narrow_x6(C.c_void_p(260), 256, p, 256, p, 64, 4096, 39, 0, None) != 0   # misaligned A
    assert L.hold_gemm_narrow_x6(p, 256, p, 256, p, 64, 0, 39, 0, None) == 0     # nothing to do


def test_wgrad_group_rejects_what_it_cannot_take():
    """the C entry point's argument checks (no GPU work is launched for a rejected list)"""
    import ctypes as C
    from hold_amd import _lib
    L = _lib.lib()
    it = (_lib.WgradItem * 3)()
    for a in it:
        a.R = a.X = a.dW = 256  # never dereferenced: every case below is rejected before the launch
        a.ldr = a.ldx = a.lddw = 256
        a.N = 256
    ws = C.c_void_p(256)
    assert L.hold_wgrad_group_x6(it, 0, 4096, ws, None) != 0           # empty list
    assert L.hold_wgrad_group_x6(it, 25, 4096, ws, None) != 0          # too many pairs
    assert L.hold_wgrad_group_x6(it, 1, 4100, ws, None) != 0           # P not a multiple of 16
    it[0].ldr = 128
    assert L.hold_wgrad_group_x6(it, 1, 4096, ws, None) != 0           # R narrower than 256 floats in memory
    it[0].ldr = 256
    it[0].N = 300
    assert L.hold_wgrad_group_x6(it, 1, 4096, ws, None) != 0           # more than 256 rows
    it[0].N = 256
    it[0].R = 260
    assert L.hold_wgrad_group_x6(it, 1, 4096, ws, None) != 0           # misaligned operand
    it[0].R = 256
    it[1].dW = 512
    assert L.hold_wgrad_group_x6(it, 3, 4096, ws, None) != 0           # pairs of one destination not adjacent


def test_fused_sdf_matches_layered_path():
    """hold_fused_sdf (LDS-resident 8-layer trunk) against the layer-by-layer GEMM path on the same weights."""
    import numpy as np
    from hold_amd import field as F, kernels as K, synthetic as syn
    dev = _dev()
    sc = syn.make_scene(2)
    sd = {k: torch.as_tensor(v).to(dev) for k, v in syn.make_state_dict(sc).items()}
    for kind, node in (("hand", "right"), ("object", "object")):
        spec = F.FieldSpec(kind)
        pre = f"nodes.{node}."
        eff = lambda p: sd[p + ".weight_v"] * (sd[p + ".weight_g"] / sd[p + ".weight_v"].norm(dim=1, keepdim=True))
        iw = [eff(pre + f"implicit_network.lin{l}") for l in range(9)]
        ib = [sd[pre + f"implicit_network.lin{l}.bias"] for l in range(9)]
        rw = [eff(pre + f"rendering_network.lin{l}") for l in range(5)]
        rb = [sd[pre + f"rendering_network.lin{l}.bias"] for l in range(5)]
        pk = F.pack_weights(spec, iw, ib, rw, rb, need_bwd=False)
        nf = F.NodeField(spec, dev)
        P = 1000
        g = torch.Generator().manual_seed(1)
        xc = torch.zeros(P, 4, device=dev)
        xc[:, :3] = (torch.rand(P, 3, generator=g) * 1.6 - 0.8).to(dev)
        barf = (torch.rand(39, generator=g).to(dev) if kind == "object" else None)
        _, h = nf._trunk(pk, xc, P, barf, keep_all=False)
        ref = torch.empty(P, 1, device=dev)
        K.rowdot(h[7], pk["w8_sdf"], 256, float(pk["b8_sdf"]), P, ref)
        out = torch.full((P, 1), 7.0, device=dev)
        wpack, bias8 = pk["fused"]
        K.fused_sdf(xc, P, wpack, bias8, pk["w8_sdf"], float(pk["b8_sdf"]), barf, out)
        err = float((out - ref).abs().max())
        assert err < 2e-5 * max(1.0, float(ref.abs().max())), (kind, err)


def test_fused_sdf_x6_matches_fp32_fused():
    """split-precision (3 bf16 limbs x 6 products) sampler trunk (limb planes in LDS, 64-point blocks, pre-split weight
    limbs) against the fp32-MFMA fused kernel"""
    from hold_amd import field as F, kernels as K, synthetic as syn
    dev = _dev()
    sc = syn.make_scene(2)
    sd = {k: torch.as_tensor(v).to(dev) for k, v in syn.make_state_dict(sc, perturb=0.05).items()}
    for kind, node in (("hand", "right"), ("object", "object")):
        spec = F.FieldSpec(kind)
        pre = f"nodes.{node}."
        eff = lambda p: sd[p + ".weight_v"] * (sd[p + ".weight_g"] / sd[p + ".weight_v"].norm(dim=1, keepdim=True))
        iw = [eff(pre + f"implicit_network.lin{l}") for l in range(9)]
        ib = [sd[pre + f"implicit_network.lin{l}.bias"] for l in range(9)]
        rw = [eff(pre + f"rendering_network.lin{l}") for l in range(5)]
        rb = [sd[pre + f"rendering_network.lin{l}.bias"] for l in range(5)]
        pk = F.pack_weights(spec, iw, ib, rw, rb, need_bwd=False)
        x6 = F.pack_x6(pk["W"][:8])
        P = 128 * 300 + 37
        g = torch.Generator().manual_seed(1)
        xc = torch.zeros(P, 4, device=dev)
        xc[:, :3] = (torch.rand(P, 3, generator=g) * 1.6 - 0.8).to(dev)
        barf = (torch.rand(39, generator=g).to(dev) if kind == "object" else None)
        wpack, bias8 = pk["fused"]
        ref = torch.empty(P, 1, device=dev)
        K.fused_sdf(xc, P, wpack, bias8, pk["w8_sdf"], float(pk["b8_sdf"]), barf, ref)
        out = torch.full((P, 1), 7.0, device=dev)
        K.fused_sdf_x6(xc, P, x6, bias8, pk["w8_sdf"], float(pk["b8_sdf"]), barf, out)
        err = float((out - ref).abs().max())
        assert err < 5e-6 * max(1.0, float(ref.abs().max())), (kind, err)


@pytest.mark.parametrize("mode", ["f16x3", "f32x6", "f32"])
@pytest.mark.parametrize("P,N,K", [(4096, 256, 256), (5000, 217, 256), (777, 256, 40), (130, 3, 256), (70000, 257, 256)])
def test_wgrad_matches_fp64_in_both_precisions(P, N, K, mode):
    """hold_wgrad (fp32 MFMA) and hold_wgrad_x6 (3-limb split, fp32 accumulate) against fp64, same tolerance"""
    import hold_amd
    from hold_amd import gemm
    dev = _dev()
    prev = hold_amd.precision()
    hold_amd.set_precision(mode)
    g = torch.Generator().manual_seed(P + N + K)
    R = torch.randn(P, N, generator=g).to(dev)
    X = torch.randn(P, K, generator=g).to(dev)
    ref = R.double().t() @ X.double()
    refb = R.double().sum(0)
    dW = torch.zeros(N, K, device=dev)
    db = torch.zeros(N, device=dev)
    try:
        gemm.wgrad(R, X, dW, db)
    finally:
        hold_amd.set_precision(prev)
    assert float((dW.double() - ref).abs().max()) < 3e-5 * max(1.0, float(ref.abs().max()))
    assert float((db.double() - refb).abs().max()) < 3e-5 * max(1.0, float(refb.abs().max()))


def test_rank1_epilogue_term_and_weighted_colsum(arith):
    """y += r1_row[p] * r1_col[n] before the epilogue function (lin8's sdf column in the input-gradient GEMM), and the
    deterministic weighted column sums that give the sdf row's weight gradient"""
    from hold_amd import gemm
    dev = _dev()
    torch.manual_seed(3)
    P, N, K = 3001, 256, 256
    A = torch.randn(P, K + 16, device=dev)[:, :K]  # a strided view (lda = K + 16) like d_rin's feature block
    W = torch.randn(N, K, device=dev) / 16
    H = torch.nn.functional.softplus(torch.randn(P, N, device=dev) * 0.05, beta=100)
    add = torch.randn(P, N, device=dev)
    row, col = torch.randn(P, device=dev), torch.randn(N, device=dev)
    out = torch.empty(P, N, device=dev)
    gemm.gemm_nt(A, W, out, epi=gemm.EPI_MUL_DSP, aux1=H, aux2=add, r1_row=row, r1_col=col)
    y = A.double() @ W.double().t() + row.double()[:, None] * col.double()[None]
    ref = y * (-torch.expm1(-100 * H.double())) + add.double()
    assert (out.double() - ref).abs().max().item() < 3e-5
    out2 = torch.empty(P, N, device=dev)
    gemm.gemm_nt(A, W, out2, r1_row=row, r1_col=col)
    assert (out2.double() - y).abs().max().item() < 3e-5
    for PP in (P, 70001, 5):
        X = torch.randn(PP, 256, device=dev)
        w = torch.randn(PP, device=dev)
        o = torch.full((256,), 2.0, device=dev)
        gemm.wcolsum(X, o, weights=w, accumulate=True)
        r = (X.double() * w.double()[:, None]).sum(0) + 2
        assert (o.double() - r).abs().max().item() < 1e-4 * max(1.0, r.abs().max().item())
        o2 = torch.empty(256, device=dev)
        gemm.wcolsum(X, o2)
        o3 = torch.empty(256, device=dev)
        gemm.wcolsum(X, o3)
        assert torch.equal(o2, o3)  # deterministic
        assert (o2.double() - X.double().sum(0)).abs().max().item() < 1e-4 * PP ** 0.5


@pytest.mark.parametrize("P,K", [(4099, 256), (1000, 128), (70000, 256), (3, 256)])
def test_colour_head_kernels(P, K):
    """hold_head3_fwd / hold_head3_bwd against torch fp64 of the same op (Linear(K, 3) + sigmoid after a ReLU layer)"""
    from hold_amd import gemm
    dev = _dev()
    g = torch.Generator().manual_seed(P + K)
    R = torch.relu(torch.randn(P, K, generator=g)).to(dev)
    W = (torch.randn(3, K, generator=g) / K ** 0.5).to(dev)
    b = torch.randn(3, generator=g).to(dev)
    out = torch.full((P, 4), -7.0, device=dev)
    gemm.head3_fwd(R, W, b, out, K=K)
    ref = torch.sigmoid(R.double() @ W.double().t() + b.double())
    assert (out[:, :3].double() - ref).abs().max().item() < 2e-6
    assert torch.all(out[:, 3] == -7.0)
    dy = torch.zeros(P, 4, device=dev)
    dy[:, :3] = torch.randn(P, 3, generator=g).to(dev)
    rr = torch.empty(P, K, device=dev)
    dW = torch.ones(3, K, device=dev)
    db4 = torch.ones(4, device=dev)
    gemm.head3_bwd(dy, R, W, rr, dW, db4, K=K, accumulate=True)
    d3 = dy[:, :3].double()
    assert (rr.double() - (d3 @ W.double()) * (R > 0)).abs().max().item() < 1e-5
    rW = d3.t() @ R.double() + 1
    assert (dW.double() - rW).abs().max().item() < 2e-5 * max(1.0, rW.abs().max().item())
    assert (db4[:3].double() - (d3.sum(0) + 1)).abs().max().item() < 1e-4 * max(1.0, P ** 0.5)
    dW2, db42 = torch.empty(3, K, device=dev), torch.empty(4, device=dev)
    gemm.head3_bwd(dy, R, W, rr, dW2, db42, K=K)
    assert (dW2.double() - (rW - 1)).abs().max().item() < 2e-5 * max(1.0, rW.abs().max().item())


@pytest.mark.parametrize("P,K,lda,ldc,epi", [(128 * 3, 256, 256, 256, "none"), (1000, 256, 256, 304, "relu"), (40000, 304, 304, 256, "relu"),
                                             (70000, 256, 256, 256, "mask"), (33, 272, 272, 256, "none"), (300000, 256, 256, 256, "relu")])
def test_gemm_r6_matches_fp64(P, K, lda, ldc, epi, split_arith):
    """hold_gemm_r6 (csrc/rgemm.hip: one 256-wide layer, register-resident) against fp64: K = 256 and the padded K = 272 / 304,
    row strides wider than the matrices, ragged P, several blocks per workgroup, the three epilogues"""
    import hold_amd
    from hold_amd import field as F, gemm
    dev = _dev()
    torch.manual_seed(P + K)
    A = torch.randn(P, lda, device=dev)
    W = torch.randn(256, K, device=dev) / 16
    b = torch.randn(256, device=dev) if epi != "mask" else None
    aux = torch.randn(P, 256, device=dev) if epi == "mask" else None
    out = torch.full((P, ldc), -7.0, device=dev)
    gemm.gemm_r6(A[:, :K] if lda == K else A, F.pack_gemm_r6(W), out, K=K, bias=b,
                 epi={"none": gemm.R6_NONE, "relu": gemm.R6_RELU, "mask": gemm.R6_MASK}[epi], aux=aux)
    y = A[:, :K].double() @ W.double().t()
    if b is not None:
        y = y + b.double()
    if epi == "relu":
        y = y.clamp_min(0)
    if epi == "mask":
        y = y * (aux > 0)
    assert float((out[:, :256].double() - y).abs().max()) < 3e-5 * max(1.0, float(y.abs().max()))
    if ldc > 256:
        assert float((out[:, 256:] + 7.0).abs().max()) == 0.0  # columns beyond the 256 outputs are not touched


@pytest.mark.parametrize("P,K,lda,ldc,epi", [(128 * 3, 256, 256, 256, "none"), (1000, 256, 256, 304, "relu"), (40000, 304, 304, 256, "relu"),
                                             (70000, 256, 256, 256, "mask"), (33, 272, 272, 256, "none"), (300000, 256, 256, 256, "relu")])
@pytest.mark.parametrize("rows", ["unit", "spread", "static"])
def test_gemm_h3_matches_fp64_and_chains_its_row_maxima(P, K, lda, ldc, epi, rows):
    """hold_gemm_h3 (csrc/rgemm_h3.hip: hold_gemm_r6 in two fp16 limbs, every operand ROW scaled by its own power of two) on the
    shapes of test_gemm_r6_matches_fp64: error against fp64 PER ROW relative to that row's own largest result (the rows' magnitudes
    spread log-uniformly over 1e-12 .. 1e+2 in 'spread': loss cotangents; a common absolute bound would only see the largest rows),
    amax_out == the exact row maxima of the result, no fallback fired; 'static' = no amax_in (the floor alone, as for lin8's input)."""
    from hold_amd import field as F, gemm, kernels as Kk
    dev = _dev()
    torch.manual_seed(P + K)
    A = torch.randn(P, lda, device=dev)
    if rows == "spread":
        A = A * (10.0 ** (torch.rand(P, 1, device=dev) * 14 - 12))
    W = torch.randn(256, K, device=dev) / 16
    b = torch.randn(256, device=dev) if epi != "mask" else None
    aux = torch.randn(P, 256, device=dev) if epi == "mask" else None
    out = torch.full((P, ldc), -7.0, device=dev)
    pk, c3 = F.pack_gemm_h3(W)
    amax_in = None if rows == "static" else A[:, :K].abs().amax(1).contiguous()
    amax_out = torch.full((P + 64,), -3.0, device=dev)
    n0 = Kk.h3_overflow_count(dev)
    gemm.gemm_h3(A[:, :K] if lda == K else A, pk, c3, out, K=K, wpack_r6=F.pack_gemm_r6(W), bias=b,
                 epi={"none": gemm.R6_NONE, "relu": gemm.R6_RELU, "mask": gemm.R6_MASK}[epi], aux=aux, amax_in=amax_in,
                 amax_floor=64.0 if rows == "static" else 0.0, amax_out=amax_out[:P])
    assert Kk.h3_overflow_count(dev) == n0
    y = A[:, :K].double() @ W.double().t()
    if b is not None:
        y = y + b.double()
    if epi == "relu":
        y = y.clamp_min(0)
    if epi == "mask":
        y = y * (aux > 0)
    # per row: against the row's own scale (|A row| max x the weights' scale, + the bias where there is one)
    den = (A[:, :K].double().abs().amax(1, keepdim=True) * float(W.abs().max()) * 16 + (float(b.abs().max()) if b is not None else 0.0)).clamp_min(1e-300)
    err = ((out[:, :256].double() - y).abs() / den).max().item()
    assert err < 3e-6, err
    if ldc > 256:
        assert float((out[:, 256:] + 7.0).abs().max()) == 0.0  # columns beyond the 256 outputs are not touched
    assert torch.equal(amax_out[:P], out[:, :256].abs().amax(1)) and float((amax_out[P:] + 3.0).abs().max()) == 0.0


def test_gemm_h3_is_bit_reproducible_whatever_ran_before():
    """GPU call 11 of round 6: rendering one frame twice gave different bits (test_full_frame_512_invariants) -- results within
    tolerance, i.e. some launch had picked another power-of-two row scale.  The scale of a row must depend on amax_in / amax_floor
    alone: the same call repeated with other kernels (NaN-filled LDS rings, other row maxima) in between must return the same bits,
    for a chain of launches as the rendering net issues it (row maxima travelling from launch to launch)."""
    import hold_amd
    from hold_amd import field as F, gemm
    dev = _dev()
    P = 128 * 256 * 3 + 77  # three blocks per workgroup and a ragged tail
    torch.manual_seed(5)
    A = torch.randn(P, 304, device=dev) * (10.0 ** (torch.rand(P, 1, device=dev) * 6 - 3))
    Ws = [torch.randn(256, 304, device=dev) / 16] + [torch.randn(256, 256, device=dev) / 16 for _ in range(2)]
    packs = [(F.pack_gemm_h3(W), F.pack_gemm_r6(W)) for W in Ws]
    b = torch.randn(256, device=dev)

    def chain():
        am = [torch.empty(P, device=dev) for _ in range(3)]
        o = [torch.empty(P, 256, device=dev) for _ in range(3)]
        am_in = A[:, :256].abs().amax(1).contiguous()
        gemm.gemm_h3(A, packs[0][0][0], packs[0][0][1], o[0], K=304, wpack_r6=packs[0][1], bias=b, epi=gemm.R6_RELU, amax_in=am_in,
                     amax_floor=float(A[:, 256:].abs().max()), amax_out=am[0])
        gemm.gemm_h3(o[0], packs[1][0][0], packs[1][0][1], o[1], K=256, wpack_r6=packs[1][1], bias=b, epi=gemm.R6_RELU, amax_in=am[0], amax_out=am[1])
        gemm.gemm_h3(o[1], packs[2][0][0], packs[2][0][1], o[2], K=256, wpack_r6=packs[2][1], epi=gemm.R6_MASK, aux=o[0], amax_in=am[1], amax_out=am[2])
        torch.cuda.synchronize()
        return o + am

    ref = chain()
    prev = hold_amd.precision()
    hold_amd.set_precision("f32x6")
    try:
        for fill in (float("nan"), 1e30, 0.0):
            R = torch.full((16 * 4096, 256), fill, device=dev)
            gemm.wgrad(R, R, torch.empty(256, 256, device=dev), None)  # 256 workgroups x a 128 KiB LDS ring of `fill`
            junk = torch.empty(P, 256, device=dev)
            gemm.gemm_h3(A * 1e3, packs[0][0][0], packs[0][0][1], junk, K=304, wpack_r6=packs[0][1], amax_floor=1e7)  # other scales in flight
            for x, y in zip(chain(), ref):
                assert torch.equal(x, y)
    finally:
        hold_amd.set_precision(prev)


def test_gemm_h3_overflow_falls_back_to_r6_with_its_row_maxima():
    """rows beyond 2^3 x the bound the caller gave (here: no amax_in, floor 64, a few rows of 1e4) leave fp16's range: the guard
    fires, the conditional hold_gemm_r6_if launch recomputes C AND amax_out, bit-identical to hold_gemm_r6, and counts the event"""
    from hold_amd import field as F, gemm, kernels as Kk
    dev = _dev()
    P, K = 40000, 256
    torch.manual_seed(3)
    A = torch.randn(P, K, device=dev)
    A[12345:12349] *= 1e4
    W = torch.randn(256, K, device=dev) / 16
    b = torch.randn(256, device=dev)
    pk, c3 = F.pack_gemm_h3(W)
    o3, o6 = torch.empty(P, 256, device=dev), torch.empty(P, 256, device=dev)
    am = torch.empty(P, device=dev)
    n0 = Kk.h3_overflow_count(dev)
    gemm.gemm_h3(A, pk, c3, o3, K=K, wpack_r6=F.pack_gemm_r6(W), bias=b, epi=gemm.R6_RELU, amax_floor=64.0, amax_out=am)
    gemm.gemm_r6(A, F.pack_gemm_r6(W), o6, K=K, bias=b, epi=gemm.R6_RELU)
    assert Kk.h3_overflow_count(dev) == n0 + 1 and Kk.h3_guard(dev)[:2].tolist() == [0, 0]
    assert torch.equal(o3, o6) and torch.equal(am, o6.abs().amax(1))
    A[12345:12349] *= 1e-4
    gemm.gemm_h3(A, pk, c3, o3, K=K, wpack_r6=F.pack_gemm_r6(W), bias=b, epi=gemm.R6_RELU, amax_floor=64.0, amax_out=am)
    assert Kk.h3_overflow_count(dev) == n0 + 1 and torch.equal(am, o3.abs().amax(1))


def _pack_bits(x):
    """[P, 256] bool -> [P, 8] int32, bit n of row p = x[p, n] (the layout of hold_gemm_h3_bits)"""
    w = (x.view(x.shape[0], 8, 32).to(torch.int64) << torch.arange(32, device=x.device)).sum(-1)
    return torch.where(w >= 2 ** 31, w - 2 ** 32, w).to(torch.int32)


@pytest.mark.parametrize("P", [128 * 5, 128 * 70 + 37, 31])
def test_gemm_h3_relu_bits_and_the_masked_launch_that_reads_them(P):
    """hold_gemm_h3_bits (round 6): a ReLU launch also writes its mask as one bit per element, and the backward's masked launch reads
    those 32 bytes per point instead of streaming the fp32 activation again -- same result, bit for bit, as masking by the
    activation itself (texture_net.py:95-101: ReLU after lin0..3; autograd multiplies the cotangent by (activation > 0))"""
    from hold_amd import field as F, gemm
    dev = _dev()
    torch.manual_seed(P)
    A = torch.randn(P, 256, device=dev)
    W1, W2 = torch.randn(256, 256, device=dev) / 16, torch.randn(256, 256, device=dev) / 16
    b = torch.randn(256, device=dev) * 0.3
    pk1, c31 = F.pack_gemm_h3(W1)
    pk2, c32 = F.pack_gemm_h3(W2)
    r, bits, am = torch.empty(P, 256, device=dev), torch.full((P, 8), 0x5a5a5a5a, dtype=torch.int32, device=dev), torch.empty(P, device=dev)
    gemm.gemm_h3(A, pk1, c31, r, K=256, wpack_r6=F.pack_gemm_r6(W1), bias=b, epi=gemm.R6_RELU, amax_in=A.abs().amax(1).contiguous(),
                 amax_out=am, bits_out=bits)
    assert 0.2 < float((r > 0).float().mean()) < 0.8
    assert torch.equal(bits, _pack_bits(r > 0))
    r_plain = torch.empty_like(r)
    gemm.gemm_h3(A, pk1, c31, r_plain, K=256, wpack_r6=F.pack_gemm_r6(W1), bias=b, epi=gemm.R6_RELU, amax_in=A.abs().amax(1).contiguous())
    assert torch.equal(r_plain, r)  # writing the bits changes nothing else
    cot = torch.randn(P, 256, device=dev) * (10.0 ** (torch.rand(P, 1, device=dev) * 8 - 6))
    am_c = cot.abs().amax(1).contiguous()
    o_aux, o_bits = torch.empty(P, 256, device=dev), torch.empty(P, 256, device=dev)
    ao_aux, ao_bits = torch.empty(P, device=dev), torch.empty(P, device=dev)
    gemm.gemm_h3(cot, pk2, c32, o_aux, K=256, wpack_r6=F.pack_gemm_r6(W2), epi=gemm.R6_MASK, aux=r, amax_in=am_c, amax_out=ao_aux)
    gemm.gemm_h3(cot, pk2, c32, o_bits, K=256, wpack_r6=F.pack_gemm_r6(W2), epi=gemm.R6_MASK, aux=r, amax_in=am_c, amax_out=ao_bits,
                 bits_in=bits)
    assert torch.equal(o_bits, o_aux) and torch.equal(ao_bits, ao_aux)
    assert torch.equal(o_bits != 0, (o_bits != 0) & (r > 0))  # (masked entries are exact zeros)


def test_gemm_h3_relu_bits_are_rebuilt_when_the_launch_falls_back():
    """an overflowing ReLU launch is recomputed by hold_gemm_r6_if, which rewrites C but knows nothing of the bits: the third
    conditional launch (relu_bits_if_kernel: fallback count moved since it last looked) rebuilds them from the recomputed C"""
    from hold_amd import field as F, gemm, kernels as Kk
    dev = _dev()
    P, K = 40000, 256
    torch.manual_seed(4)
    A = torch.randn(P, K, device=dev)
    A[2345:2349] *= 1e4
    W = torch.randn(256, K, device=dev) / 16
    b = torch.randn(256, device=dev)
    pk, c3 = F.pack_gemm_h3(W)
    o3, o6 = torch.empty(P, 256, device=dev), torch.empty(P, 256, device=dev)
    bits = torch.zeros(P, 8, dtype=torch.int32, device=dev)
    n0 = Kk.h3_overflow_count(dev)
    gemm.gemm_h3(A, pk, c3, o3, K=K, wpack_r6=F.pack_gemm_r6(W), bias=b, epi=gemm.R6_RELU, amax_floor=64.0, bits_out=bits)
    gemm.gemm_r6(A, F.pack_gemm_r6(W), o6, K=K, bias=b, epi=gemm.R6_RELU)
    g = Kk.h3_guard(dev).tolist()
    assert Kk.h3_overflow_count(dev) == n0 + 1 and g[:2] == [0, 0] and g[3] == g[2]
    assert torch.equal(o3, o6) and torch.equal(bits, _pack_bits(o6 > 0))
    A[2345:2349] *= 1e-4  # no overflow: the bits come from the kernel's own registers, the third launch exits at once
    bits.fill_(-1)
    gemm.gemm_h3(A, pk, c3, o3, K=K, wpack_r6=F.pack_gemm_r6(W), bias=b, epi=gemm.R6_RELU, amax_floor=64.0, bits_out=bits)
    assert Kk.h3_overflow_count(dev) == n0 + 1 and torch.equal(bits, _pack_bits(o3 > 0))


@pytest.mark.parametrize("rscale,xscale", [(1e-9, 1.0), (3e-7, 40.0), (1e4, 1e-5), (1.0, 1.0)])
def test_wgrad_h3_scales_follow_the_operands(rscale, xscale):
    """the two-limb fp16 weight gradient picks power-of-two operand scales per workgroup from a sample of its rows: loss
    cotangents of 1e-9, activations of tens, and operands whose magnitude CHANGES along the points (every workgroup has its
    own scales) must come out at the relative accuracy of the unit-scale case; rows the sample did not see may exceed the
    sampled maximum by up to 2^9 without overflow (here: an isolated 200 x outlier)."""
    import hold_amd
    from hold_amd import gemm
    dev = _dev()
    prev = hold_amd.precision()
    hold_amd.set_precision("f16x3")
    try:
        P = 16 * 65536  # 256 workgroups x 4 096 rows: the scale sample reads rows 0..3 of every 64
        torch.manual_seed(7)
        ramp = torch.logspace(0, 3, P, device=dev).view(P, 1)  # magnitudes grow 1000 x along the points
        R = torch.randn(P, 256, device=dev) * rscale * ramp
        X = torch.randn(P, 256, device=dev) * xscale / ramp.flip(0)
        # an outlier no sample row contains (row 12345 = row 57 of its workgroup's rows 12288..16383; 57 % 64 >= 4): 200 x the
        # largest value of that workgroup's rows
        R[12345, 17] = 200.0 * float(R[12288:16384].abs().max())
        dW = torch.zeros(256, 256, device=dev)
        db = torch.zeros(256, device=dev)
        gemm.wgrad(R, X, dW, db)
        ref = R.double().t() @ X.double()
        assert torch.isfinite(dW).all()
        assert ((dW.double() - ref).abs().max() / ref.abs().max()).item() < 1e-5
        # the same bound row by row of dW (a row = one column of R: the outlier's row has a far larger norm than the rest)
        rel = (dW.double() - ref).abs().max(1).values / ref.abs().max(1).values
        assert rel.max().item() < 3e-5, rel.max().item()
        refb = R.double().sum(0)
        assert ((db.double() - refb).abs().max() / refb.abs().max()).item() < 1e-5
        # all-zero operands: scale 1, result exactly zero
        gemm.wgrad(torch.zeros_like(R), X, dW, None)
        assert float(dW.abs().max()) == 0.0
        # HEAVY TAILS: rows the sample does not see 1e6 x larger than every sampled row (2^9 is the sampled scale's headroom):
        # the workgroups that own them notice the overflow in their exact running maxima and repeat their share with exact
        # scales -- the result is finite and as accurate as before (round 5, GPU call 12: the version without the retry
        # returned NaN gradients on a sharp-density scene, whose cotangents look like this)
        R2 = torch.randn(P, 256, device=dev) * rscale
        rows = torch.arange(P, device=dev)
        hidden = (rows % 64 >= 4) & (rows % 977 == 5)  # never in a sampled group of four rows
        R2[hidden] *= 1e6
        dW2 = torch.zeros(256, 256, device=dev)
        gemm.wgrad(R2, X, dW2, None)
        ref2 = R2.double().t() @ X.double()
        assert torch.isfinite(dW2).all()
        assert ((dW2.double() - ref2).abs().max() / ref2.abs().max()).item() < 1e-5
        # UNDER-scaling (advisor, round 5): every row the sample sees is exactly zero (rays that miss the node, alpha == 0), so
        # the sampled scale is 1, while the rows it does not see carry cotangents of ~1e-6 (and one workgroup's of ~1e-9): split
        # at scale 1 their lo limbs would be fp16 subnormals and anything below 3e-8 would vanish.  The exact running maximum
        # tells the workgroup (non-zero, far below the sample's target) and it repeats its share with exact scales.
        R3 = torch.randn(P, 256, device=dev) * 1e-6 * rscale
        R3[rows % 64 < 4] = 0.0
        R3[8192:12288] *= 1e-3
        dW3 = torch.zeros(256, 256, device=dev)
        db3 = torch.zeros(256, device=dev)
        gemm.wgrad(R3, X, dW3, db3)
        ref3 = R3.double().t() @ X.double()
        assert ((dW3.double() - ref3).abs().max() / ref3.abs().max()).item() < 1e-5
        # the 1e-9 workgroup on its own (its share is invisible in the full sum): same relative accuracy
        dW4 = torch.zeros(256, 256, device=dev)
        gemm.wgrad(R3[8192:12288], X[8192:12288], dW4, None)
        ref4 = R3[8192:12288].double().t() @ X[8192:12288].double()
        assert ((dW4.double() - ref4).abs().max() / ref4.abs().max()).item() < 1e-5
        refb3 = R3.double().sum(0)
        assert ((db3.double() - refb3).abs().max() / refb3.abs().max()).item() < 1e-5
    finally:
        hold_amd.set_precision(prev)
