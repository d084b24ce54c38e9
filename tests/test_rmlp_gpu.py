"""GPU parity of the register-resident trunk (hold_fused_sdf_r6 / hold_trunk_r6, csrc/rmlp.hip) against a torch fp64
restatement of ImplicitNet.forward's lin0..lin7 (shape_net.py:84-130) and against the LDS-resident kernels it replaces."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu

SK = 217


def _dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _sp(y):
    return torch.nn.functional.softplus(y, beta=100)


def _embed(x, barf=None):
    cols = [x]
    for k in range(6):
        cols += [torch.sin(x * 2.0 ** k), torch.cos(x * 2.0 ** k)]
    e = torch.cat(cols, 1)
    return e if barf is None else e * barf


def _net(seed, dev, barf):
    g = torch.Generator().manual_seed(seed)
    w0 = torch.zeros(256, 40)
    w0[:, :39] = torch.randn(256, 39, generator=g) / 6
    S = torch.randn(7, 256, 256, generator=g) / 16
    S[2, SK:] = 0
    bias = torch.randn(8, 256, generator=g) * 0.05
    bias[3, SK:] = 0
    w8 = torch.randn(256, generator=g) / 16
    bw = (torch.rand(39, generator=g) if barf else None)
    to = lambda t: None if t is None else t.to(dev)
    return to(w0), to(S), to(bias), to(w8), to(bw)


def _ref(x, w0, S, bias, bw):
    """fp64 trunk: list of h_0..h_7 with h_3 = [h3 (217) | embedding (39)]"""
    emb = _embed(x.double(), None if bw is None else bw.double())
    hs, cur = [], emb
    for l in range(8):
        W = w0.double()[:, :39] if l == 0 else S[l - 1].double()
        cur = _sp(cur @ W.t() + bias[l].double())
        if l == 3:
            cur = torch.cat([cur[:, :SK], emb], 1)
        hs.append(cur)
    return hs


@pytest.mark.parametrize("barf", [False, True])
@pytest.mark.parametrize("P", [1, 33, 130, 1000, 128 * 300 + 77])
def test_fused_sdf_r6_matches_fp64_and_x6(P, barf):
    from hold_amd import field as F, kernels as K
    dev = _dev()
    w0, S, bias, w8, bw = _net(P, dev, barf)
    g = torch.Generator().manual_seed(P + 1)
    xc = torch.zeros(P, 4)
    xc[:, :3] = torch.rand(P, 3, generator=g) * 1.6 - 0.8
    xc = xc.to(dev)
    big = torch.full((P + 130, 1), 9.0, device=dev)
    out = big[:P]
    K.fused_sdf_r6(xc, P, F.pack_r6(w0, S), bias, w8, torch.full((1,), 0.25, device=xc.device), bw, out)
    torch.cuda.synchronize()
    assert torch.all(big[P:] == 9.0)
    hs = _ref(xc[:, :3], w0, S, bias, bw)
    ref = hs[7] @ w8.double() + 0.25
    err = (out[:, 0].double() - ref).abs().max().item()
    assert err < 3e-5 * max(1.0, ref.abs().max().item()), err
    out2 = torch.empty(P, 1, device=dev)
    K.fused_sdf_x6(xc, P, torch.cat([F.pack_x6([w0]), F.pack_x6_stack(S)]), bias, w8, 0.25, bw, out2)
    assert (out - out2).abs().max().item() < 2e-5


@pytest.mark.parametrize("barf", [False, True])
@pytest.mark.parametrize("P", [1, 130, 1000, 128 * 257 + 3])
def test_trunk_r6_stores_every_layer(P, barf):
    from hold_amd import field as F, kernels as K
    dev = _dev()
    w0, S, bias, w8, bw = _net(P + 7, dev, barf)
    g = torch.Generator().manual_seed(P + 2)
    xc = torch.zeros(P, 4)
    xc[:, :3] = torch.rand(P, 3, generator=g) * 1.6 - 0.8
    xc = xc.to(dev)
    big = [torch.full((P + 130, 256), 9.0, device=dev) for _ in range(8)]
    h = [b[:P] for b in big]
    K.trunk_r6(xc, P, F.pack_r6(w0, S), bias, bw, h)
    torch.cuda.synchronize()
    for b in big:
        assert torch.all(b[P:] == 9.0)  # rows >= P are never written (buffer range check)
    emb = _embed(xc[:, :3].double(), None if bw is None else bw.double())
    cur = emb
    for l in range(8):
        W = w0.double()[:, :39] if l == 0 else S[l - 1].double()
        cur = _sp(cur @ W.t() + bias[l].double())
        if l == 3:
            cur = torch.cat([cur[:, :SK], emb], 1)
        err = (h[l].double() - cur).abs().max().item()
        assert err < 3e-5 * max(1.0, cur.abs().max().item()), (l, err)
        # small activations keep RELATIVE accuracy (the backward sweeps recover softplus' from the stored h): for y < 0
        # softplus ~ exp(100 y) / 100, so the fp32 rounding of the pre-activation (1e-6 absolute) alone is 1e-4 relative;
        # a log1p that rounded e away (log(1 + e) at e <= 1e-3) would show up as 1e-2 and more
        small = (cur < 1e-4) & (cur > 1e-30)  # below that exp2 flushes to zero: absolute error 1e-32
        if small.any() and l != 3:
            rel = ((h[l].double() - cur).abs() / cur)[small].max().item()
            assert rel < 1e-3, (l, rel)
        cur = h[l].double()  # follow the kernel's own rounding from layer to layer


# ---- the two-limb fp16 arithmetic "f16x3" (csrc/rmlp_h3.hip) -------------------------------------------------------------
def _h3_pack(w0, S, bias):
    from hold_amd import field as F
    pk, sw = F.pack_h3(w0, S)
    return pk, (bias * (sw * F.H3_ACT_SCALE).view(8, 1)).contiguous(), (1.0 / sw).contiguous()


@pytest.mark.parametrize("barf", [False, True])
@pytest.mark.parametrize("P", [1, 33, 130, 1000, 128 * 300 + 77])
def test_fused_sdf_h3_matches_fp64_and_r6(P, barf):
    """the contract and tolerances of test_fused_sdf_r6_matches_fp64_and_x6 for hold_fused_sdf_h3"""
    from hold_amd import field as F, kernels as K
    dev = _dev()
    w0, S, bias, w8, bw = _net(P, dev, barf)
    g = torch.Generator().manual_seed(P + 1)
    xc = torch.zeros(P, 4)
    xc[:, :3] = torch.rand(P, 3, generator=g) * 1.6 - 0.8
    xc = xc.to(dev)
    big = torch.full((P + 130, 1), 9.0, device=dev)
    out = big[:P]
    pk, bs, c3 = _h3_pack(w0, S, bias)
    b8 = torch.full((1,), 0.25, device=dev)
    K.fused_sdf_h3(xc, P, pk, bs, c3, w8, b8, bw, out)
    torch.cuda.synchronize()
    assert torch.all(big[P:] == 9.0)
    hs = _ref(xc[:, :3], w0, S, bias, bw)
    ref = hs[7] @ w8.double() + 0.25
    err = (out[:, 0].double() - ref).abs().max().item()
    assert err < 3e-5 * max(1.0, ref.abs().max().item()), err
    out2 = torch.empty(P, 1, device=dev)
    K.fused_sdf_r6(xc, P, F.pack_r6(w0, S), bias, w8, b8, bw, out2)
    assert (out - out2).abs().max().item() < 2e-5


@pytest.mark.parametrize("barf", [False, True])
@pytest.mark.parametrize("P", [1, 130, 1000, 128 * 257 + 3])
def test_trunk_h3_stores_every_layer(P, barf):
    """the contract and tolerances of test_trunk_r6_stores_every_layer for hold_trunk_h3 (h in fp32, unscaled)"""
    from hold_amd import kernels as K
    dev = _dev()
    w0, S, bias, w8, bw = _net(P + 7, dev, barf)
    g = torch.Generator().manual_seed(P + 2)
    xc = torch.zeros(P, 4)
    xc[:, :3] = torch.rand(P, 3, generator=g) * 1.6 - 0.8
    xc = xc.to(dev)
    big = [torch.full((P + 130, 256), 9.0, device=dev) for _ in range(8)]
    h = [b[:P] for b in big]
    pk, bs, c3 = _h3_pack(w0, S, bias)
    K.trunk_h3(xc, P, pk, bs, c3, bw, h)
    torch.cuda.synchronize()
    for b in big:
        assert torch.all(b[P:] == 9.0)
    emb = _embed(xc[:, :3].double(), None if bw is None else bw.double())
    cur = emb
    for l in range(8):
        W = w0.double()[:, :39] if l == 0 else S[l - 1].double()
        cur = _sp(cur @ W.t() + bias[l].double())
        if l == 3:
            cur = torch.cat([cur[:, :SK], emb], 1)
        err = (h[l].double() - cur).abs().max().item()
        assert err < 3e-5 * max(1.0, cur.abs().max().item()), (l, err)
        small = (cur < 1e-4) & (cur > 1e-30)
        if small.any() and l != 3:
            rel = ((h[l].double() - cur).abs() / cur)[small].max().item()
            assert rel < 1e-3, (l, rel)
        cur = h[l].double()


def test_h3_error_against_fp64_is_within_one_and_a_half_of_r6_on_half_a_million_points():
    """the gate of the f16x3 arithmetic: on 524 288 points the max abs error of the sdf against an fp64 trunk is
    at most 1.5 x that of the exact three-limb bf16 kernel (hold_fused_sdf_r6) -- for a geometric-initialisation-like net
    (the scene's) and for the random net of the tests above; rms errors are printed for the record (profiles/r05_h3_gate.json
    is written by scripts/h3_gate.py with the same code path)."""
    from hold_amd import field as F, kernels as K
    dev = _dev()
    P = 524288
    for seed, scale in ((5, 1.0), (6, 3.0)):
        w0, S, bias, w8, bw = _net(seed, dev, False)
        S = S * scale  # larger weights: activations up to a few units, several binades of dynamic range per layer
        g = torch.Generator().manual_seed(seed + 1)
        xc = torch.zeros(P, 4)
        xc[:, :3] = torch.rand(P, 3, generator=g) * 1.6 - 0.8
        xc = xc.to(dev)
        b8 = torch.full((1,), 0.25, device=dev)
        o6, o3 = torch.empty(P, 1, device=dev), torch.empty(P, 1, device=dev)
        K.fused_sdf_r6(xc, P, F.pack_r6(w0, S), bias, w8, b8, bw, o6)
        pk, bs, c3 = _h3_pack(w0, S, bias)
        K.fused_sdf_h3(xc, P, pk, bs, c3, w8, b8, bw, o3)
        ref = torch.cat([(_ref(xc[i:i + 65536, :3], w0, S, bias, bw)[7] @ w8.double() + 0.25) for i in range(0, P, 65536)])
        e6, e3 = (o6[:, 0].double() - ref).abs(), (o3[:, 0].double() - ref).abs()
        print(f"seed {seed}: |sdf| max {ref.abs().max().item():.3f}; r6 max {e6.max().item():.3e} rms {e6.pow(2).mean().sqrt().item():.3e}; "
              f"h3 max {e3.max().item():.3e} rms {e3.pow(2).mean().sqrt().item():.3e}")
        assert e3.max().item() <= 1.5 * e6.max().item(), (seed, e3.max().item(), e6.max().item())
        assert e3.pow(2).mean().sqrt().item() <= 1.5 * e6.pow(2).mean().sqrt().item()


def test_h3_kernels_do_not_depend_on_stale_lds():
    """round 5, GPU calls 4 / 5: the padded fourth k step of layer 0 read LDS bytes no one had written; zero weights times a
    stale inf / NaN bit pattern made every accumulator NaN -- invisible to the tests above, which start from a quiet LDS.
    Here every compute unit's LDS is filled with NaN first (the whole-dW weight gradient stages its operand rows in a
    128 KiB ring: all-NaN operands on 256 workgroups), then the h3 kernels must return bit for bit what they return
    after a zero fill."""
    import hold_amd
    from hold_amd import gemm, kernels as K
    dev = _dev()
    P = 128 * 256 * 2
    w0, S, bias, w8, bw = _net(3, dev, True)
    g = torch.Generator().manual_seed(9)
    xc = torch.zeros(P, 4)
    xc[:, :3] = torch.rand(P, 3, generator=g) * 1.6 - 0.8
    xc = xc.to(dev)
    pk, bs, c3 = _h3_pack(w0, S, bias)
    b8 = torch.full((1,), 0.25, device=dev)
    prev = hold_amd.precision()
    hold_amd.set_precision("f32x6")
    try:
        res = []
        for fill in (0.0, float("nan"), float("inf")):
            R = torch.full((16 * 4096, 256), fill, device=dev)
            dW = torch.empty(256, 256, device=dev)
            gemm.wgrad(R, R, dW, None)  # 256 workgroups x a 128 KiB LDS ring of `fill`
            out = torch.empty(P, 1, device=dev)
            K.fused_sdf_h3(xc, P, pk, bs, c3, w8, b8, bw, out)
            gemm.wgrad(R, R, dW, None)
            h = [torch.empty(P, 256, device=dev) for _ in range(8)]
            K.trunk_h3(xc, P, pk, bs, c3, bw, h)
            torch.cuda.synchronize()
            res.append((out, h))
        for out, h in res[1:]:
            assert torch.isfinite(out).all()
            assert torch.equal(out, res[0][0])
            for a, b in zip(h, res[0][1]):
                assert torch.equal(a, b)
    finally:
        hold_amd.set_precision(prev)


@pytest.mark.parametrize("where", ["activation", "coordinate"])
def test_h3_overflow_is_caught_on_the_device_and_recomputed_in_f32x6(where):
    """the f16x3 trunk scales its activations by the constant 2^6, so an activation >= 1023.5 has no fp16
    representation -- and an fp16 infinity is not guaranteed to stay visible (inf x negative weight -> -inf -> softplus -> 0).
    The kernels keep the exact maximum of everything they split; a launch in which one left the range sets the guard word, and the
    conditional f32x6 launch the entry point enqueues behind it recomputes the WHOLE launch: the result is hold_fused_sdf_r6's /
    hold_trunk_r6's bit for bit, the event is counted, the guard is armed again -- and a launch without an overflow is untouched by
    all of it.  Driven here by one layer-1 bias of 1100 (a softplus output beyond 1023.5 for every point) or by ONE point whose
    coordinate is 2000 (the raw coordinates are activations of layer 0)."""
    from hold_amd import field as F, kernels as K
    dev = _dev()
    P = 128 * 300 + 77
    w0, S, bias, w8, bw = _net(11, dev, True)
    g = torch.Generator().manual_seed(12)
    xc = torch.zeros(P, 4)
    xc[:, :3] = torch.rand(P, 3, generator=g) * 1.6 - 0.8
    bias_ovf, xc_ovf = bias.clone(), xc.clone()
    if where == "activation":
        bias_ovf[1, 5] = 1100.0
    else:
        xc_ovf[12345, 1] = -2000.0  # a NEGATIVE coordinate: the guard looks at magnitudes
        bw = bw.clone()
        bw[:3] = 1.0  # the raw coordinates' embedding weight (BARF only ever weights the frequencies, embedders.py:92-105)
    xc, xc_ovf = xc.to(dev), xc_ovf.to(dev)
    b8 = torch.full((1,), 0.25, device=dev)
    guard = K.h3_guard(dev)
    pr6 = F.pack_r6(w0, S)

    def both(xq, bq):
        pk, bs, c3 = _h3_pack(w0, S, bq)
        o3, o6 = torch.empty(P, 1, device=dev), torch.empty(P, 1, device=dev)
        K.fused_sdf_h3(xq, P, pk, bs, c3, w8, b8, bw, o3, pr6, bq)
        K.fused_sdf_r6(xq, P, pr6, bq, w8, b8, bw, o6)
        h3 = [torch.empty(P, 256, device=dev) for _ in range(8)]
        h6 = [torch.empty(P, 256, device=dev) for _ in range(8)]
        K.trunk_h3(xq, P, pk, bs, c3, bw, h3, pr6, bq)
        K.trunk_r6(xq, P, pr6, bq, bw, h6)
        return o3, o6, h3, h6

    n0 = K.h3_overflow_count(dev)
    o3, o6, h3, h6 = both(xc, bias)  # no overflow: the f16x3 results, not the fallback's
    assert K.h3_overflow_count(dev) == n0 and guard[:2].tolist() == [0, 0]
    assert not torch.equal(o3, o6) and (o3 - o6).abs().max().item() < 2e-5
    o3, o6, h3, h6 = both(xc_ovf, bias_ovf)
    assert K.h3_overflow_count(dev) == n0 + 2 and guard[:2].tolist() == [0, 0]
    assert torch.equal(o3, o6)
    for a, b in zip(h3, h6):
        assert torch.equal(a, b)
    if where == "activation":
        assert float(h6[1][:, 5].min()) > 1023.5 and torch.isfinite(o6).all()
    # without the fallback operands the flag is the report: it stays set until the caller clears it
    pk, bs, c3 = _h3_pack(w0, S, bias_ovf)
    K.fused_sdf_h3(xc_ovf, P, pk, bs, c3, w8, b8, bw, o3)
    assert guard[:3].tolist() == [1, 0, n0 + 2]
    guard[0] = 0
    # ... and the next clean launch is a clean launch
    o3b, _, _, _ = both(xc, bias)
    assert K.h3_overflow_count(dev) == n0 + 2 and guard[:2].tolist() == [0, 0]
    pk, bs, c3 = _h3_pack(w0, S, bias)
    o3c = torch.empty(P, 1, device=dev)
    K.fused_sdf_h3(xc, P, pk, bs, c3, w8, b8, bw, o3c)
    assert torch.equal(o3b, o3c)
