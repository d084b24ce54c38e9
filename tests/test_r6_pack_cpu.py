"""Index algebra of the register-resident trunk (hold_amd/csrc/rmlp.hip) checked on the CPU: a lane-level numpy model of
v_mfma_f32_32x32x16_bf16 (operand / accumulator register layout of /opt/skills/guides/cdna_hip_programming.md section 3)
runs the kernel's dataflow -- the previous layer's accumulator registers 8 q .. 8 q + 7 of tile nt ARE the B operand of
k step j = 2 nt + q -- on the weight stream produced by hold_amd.field.pack_r6, and must reproduce a plain
ImplicitNet trunk (shape_net.py:84-130: softplus(beta = 100), skip concat at layer 4).  This pins pack_r6 / r6_kmap and
the feature <-> register formulas the kernel uses; the arithmetic itself is tested on the GPU (test_rmlp_gpu.py)."""
import numpy as np
import torch

from hold_amd.field import pack_r6

NSTEP, L0S, LKS, SKIP = 115, 3, 16, 217


def mfma_32x32x16(A_lane, B_lane, acc):
    """A_lane / B_lane [64 lanes][8]: lane l holds A[i = l % 32][k = 8 (l // 32) + e] / B[k = 8 (l // 32) + e][j = l % 32];
    acc [16 regs][64 lanes]: lane l holds D[i = 8 g + 4 (l // 32) + r][j = l % 32] in register 4 g + r."""
    A = np.zeros((32, 16))
    B = np.zeros((16, 32))
    for l in range(64):
        A[l % 32, 8 * (l // 32):8 * (l // 32) + 8] = A_lane[l]
        B[8 * (l // 32):8 * (l // 32) + 8, l % 32] = B_lane[l]
    D = A @ B
    for l in range(64):
        for g in range(4):
            for r in range(4):
                acc[4 * g + r, l] += D[8 * g + 4 * (l // 32) + r, l % 32]


def softplus100(y):
    return np.maximum(y, 0) + np.log1p(np.exp(-np.abs(100 * y))) / 100


def test_r6_stream_reproduces_the_trunk():
    g = torch.Generator().manual_seed(3)
    w0 = torch.randn(256, 40, generator=g) * 0.2
    w0[:, 39] = 0
    S = torch.randn(7, 256, 256, generator=g) * 0.08
    S[2, SKIP:] = 0  # layer 3 has 217 outputs (rows zero-padded)
    bias = torch.randn(8, 256, generator=g) * 0.05
    bias[3, SKIP:] = 0
    emb = torch.randn(32, 48, generator=g)
    emb[:, 39:] = 0
    pack = pack_r6(w0, S).float().numpy().reshape(NSTEP, 8, 3, 2, 32, 8).astype(np.float64)
    Wsum = pack.sum(2)  # limbs add up exactly: [step][nt][h][i][e]
    A_frag = Wsum.reshape(NSTEP, 8, 64, 8)  # lane = 32 h + i
    b = bias.numpy().astype(np.float64)
    x = emb.numpy().astype(np.float64)
    lanes = np.arange(64)
    hh, li = lanes // 32, lanes % 32

    def init(layer):
        acc = np.zeros((8, 16, 64))
        for nt in range(8):
            for gq in range(4):
                for r in range(4):
                    acc[nt, 4 * gq + r] = b[layer, 32 * nt + 8 * gq + 4 * hh + r]
        return acc

    # layer 0: B limbs from the embedding in natural k order
    Q = init(0)
    for j in range(L0S):
        Bl = np.stack([x[li[l], 16 * j + 8 * hh[l]:16 * j + 8 * hh[l] + 8] for l in range(64)])
        for nt in range(8):
            mfma_32x32x16(A_frag[j, nt], Bl, Q[nt])
    for layer in range(1, 8):
        P, Q = Q, init(layer)
        t0 = L0S + (layer - 1) * LKS
        for j in range(LKS):
            nt_in, q = j // 2, j % 2
            Bl = softplus100(P[nt_in, 8 * q:8 * q + 8, :]).T.copy()  # [lane][e]
            if layer == 4:  # kernel: value e of lane half hh is feature f; columns 217.. come from the embedding
                for l in range(64):
                    for e in range(8):
                        f = 32 * nt_in + 16 * q + 8 * (e // 4) + 4 * hh[l] + e % 4
                        if f >= SKIP:
                            Bl[l, e] = x[li[l], f - SKIP]
            for nt in range(8):
                mfma_32x32x16(A_frag[t0 + j, nt], Bl, Q[nt])
    h7 = np.zeros((32, 256))
    for nt in range(8):
        for gq in range(4):
            for r in range(4):
                for l in range(64):
                    h7[li[l], 32 * nt + 8 * gq + 4 * hh[l] + r] = softplus100(Q[nt, 4 * gq + r, l])

    # plain trunk
    W = [w0.numpy().astype(np.float64)] + [S[i].numpy().astype(np.float64) for i in range(7)]
    a = x[:, :40]
    for layer in range(8):
        if layer == 4:
            a = np.concatenate([a[:, :SKIP], x[:, :39]], 1)
        a = softplus100(a @ W[layer].T[:a.shape[1]] + b[layer])
    assert np.abs(h7 - a).max() < 1e-9, np.abs(h7 - a).max()


def test_h3_stream_is_the_r6_stream_in_two_fp16_limbs():
    """hold_trunk_h3's weight stream (field.pack_h3, csrc/rmlp_h3.hip): the k order / tiling of pack_r6 (pinned to the lane
    model above) with TWO fp16 limbs of the scaled weights: hi + lo = s_w W to 2^-22 relative (2^-25 absolute below the fp16
    normal range of lo), s_w[l] an exact power of two with max |W_l| s_w in [2^13, 2^14); and the pack that
    field.pack_weights gathers from the flat source in mode f16x3 is the same stream bit for bit."""
    import hold_amd
    from hold_amd import field as F
    g = torch.Generator().manual_seed(11)
    w0 = torch.randn(256, 40, generator=g) * 0.2
    w0[:, 39] = 0
    S = torch.randn(7, 256, 256, generator=g) * torch.tensor([0.08, 0.3, 0.001, 0.08, 2.0, 0.08, 0.08]).view(7, 1, 1)
    S[2, SKIP:] = 0
    pk, sw = F.pack_h3(w0, S)
    NS3 = 116  # the f16x3 stream pads layer 0 to four k steps (K = 64)
    assert pk.dtype == torch.float16 and pk.numel() * 2 == NS3 * 16 * 1024
    mant, ex = torch.frexp(sw)
    assert torch.all(mant == 0.5)  # exact powers of two
    amax = torch.cat([w0.abs().amax().view(1), S.abs().amax(dim=(1, 2))])
    assert torch.all(amax * sw >= 2.0 ** 13) and torch.all(amax * sw < 2.0 ** 14)
    h3 = pk.double().reshape(NS3, 8, 2, 2, 32, 8)
    assert float(h3[L0S].abs().max()) == 0.0  # the padding k step of layer 0: zero weights
    h3 = torch.cat([h3[:L0S], h3[L0S + 1:]])  # ... the other 115 steps are pack_r6's
    r6 = pack_r6(w0, S).double().reshape(NSTEP, 8, 3, 2, 32, 8).sum(2)  # exact fp32 weights in the stream's order
    layer = torch.cat([torch.zeros(L0S, dtype=torch.long), 1 + torch.arange(7).repeat_interleave(LKS)])
    scale = sw.double()[layer].view(NSTEP, 1, 1, 1, 1)
    ws = r6 * scale
    err = (h3.sum(2) - ws).abs()
    assert torch.all(err <= ws.abs() * 2.0 ** -22 + 2.0 ** -25), float((err - ws.abs() * 2.0 ** -22).max())
    hi = h3[:, :, 0]
    assert torch.equal(hi, ws.to(torch.float32).to(torch.float16).double())  # limb 0 = RN_f16 of the scaled weight
    # the flat-source gather of pack_weights
    prev = hold_amd.precision()
    hold_amd.set_precision("f16x3")
    try:
        spec = F.FieldSpec("object")
        iw = [w0[:, :39].contiguous()] + [S[l].clone() for l in range(7)] + [torch.randn(257, 256, generator=g) * 0.05]
        iw[3] = iw[3][:SKIP].contiguous()
        ib = [torch.randn(w.shape[0], generator=g) * 0.1 for w in iw]
        pkw = F.pack_weights(spec, iw, ib, None, None, need_bwd=False)
        S2 = S.clone()
        S2[3] = S2[3] / 2 ** 0.5  # pack_weights folds the skip concat's 1 / sqrt(2) into lin4
        ref, sw2 = F.pack_h3(w0, S2)
        assert torch.equal(pkw["trunk_h3"], ref)
        assert torch.equal(pkw["c3_h3"], 1.0 / sw2)
        b8 = torch.stack([torch.nn.functional.pad(b, (0, 256 - b.shape[0])) for b in ib[:8]])
        assert torch.equal(pkw["bias8_h3"], b8 * (sw2 * F.H3_ACT_SCALE).view(8, 1))
        assert "trunk_r6" in pkw  # the f32x6 fallback of every f16x3 kernel still finds its bf16 stream
        # the descending sweeps' stream (hold_chain_h3, DSP): chain layer j = W_{7-j}^T with THAT matrix's scale -- the same limbs,
        # gathered transposed -- and c3 in chain order; equal to the stand-alone packer on the transposed matrices
        MT = S2.transpose(1, 2).flip(0).contiguous()
        refb, swb = F.pack_h3_stack(MT)
        assert torch.equal(swb, sw2[1:].flip(0))  # a matrix and its transpose share their maximum
        assert torch.equal(pkw["chain_bwd_h3"], refb)
        assert torch.equal(pkw["c3_bwd_h3"], 1.0 / swb)
        # ... which is pack_r6_stack's stream in two fp16 limbs (rows and k order pinned to the lane model above)
        r6b = F.pack_r6_stack(MT).double().reshape(7 * 16, 8, 3, 2, 32, 8).sum(2)
        h3b = refb.double().reshape(7 * 16, 8, 2, 2, 32, 8)
        wsb = r6b * swb.double().repeat_interleave(16).view(-1, 1, 1, 1, 1)
        errb = (h3b.sum(2) - wsb).abs()
        assert torch.all(errb <= wsb.abs() * 2.0 ** -22 + 2.0 ** -25)
    finally:
        hold_amd.set_precision(prev)


def test_h3_schedule_tables_are_permutations_that_respect_the_dependencies():
    """hold_amd/csrc/rmlp_h3_sched.h (generated, committed) == what scripts/gen_h3_schedule.py prints; every table is a
    permutation of the kernel's micro-operation indices in which an operation follows the ones it reads"""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "scripts", "gen_h3_schedule.py")], capture_output=True, text=True,
                         check=True).stdout
    assert out == open(os.path.join(root, "hold_amd", "csrc", "rmlp_h3_sched.h")).read()
    sys.path.insert(0, os.path.join(root, "scripts"))
    import gen_h3_schedule as G
    import re
    for head, name in ((True, "H3_SCHED_HEAD"), (False, "H3_SCHED_STORE")):
        order = [int(v) for v in re.search(r"#define %s_ORDER \{([^}]*)\}" % name, out).group(1).split(",")]
        end = [int(v) for v in re.search(r"#define %s_END \{([^}]*)\}" % name, out).group(1).split(",")]
        ops, _ = G.build(head)
        assert sorted(order) == sorted(ops) == list(range(len(order)))
        pos = {k: i for i, k in enumerate(order)}
        assert all(pos[d] < pos[k] for k, o in ops.items() for d in o["deps"])
        # v_exp / v_log are opaque asm to the hazard recogniser: their consumers must not be the very next instruction
        assert all(pos[k] - pos[d] >= 2 for k, o in ops.items() for d in o["deps"] if ops[d]["trans"])
        assert len(end) == 24 and end[-1] == len(order) and all(a <= b for a, b in zip(end, end[1:]))
        assert max(b - a for a, b in zip([0] + end, end)) <= 12  # the kernel's per-gap loop bound


def test_gemm_r6_stream_reproduces_a_layer():
    """hold_gemm_r6 (csrc/rgemm.hip) on the same lane-level model: the B operand of k step e is the lane's own row, columns
    16 e + 8 (i / 4) + 4 hh + i % 4 (two 16-byte pieces at 16 e + 4 hh and 16 e + 8 + 4 hh); with the weight stream of
    field.pack_gemm_r6 the accumulators must hold A . W^T + b in the layout the stores assume (register 4 g + r of tile nt =
    output 32 nt + 8 g + 4 hh + r), for K = 256 and for the zero-padded K = 304 (20 k steps, padded steps re-read k step 0)."""
    from hold_amd.field import pack_gemm_r6
    g = torch.Generator().manual_seed(5)
    lanes = np.arange(64)
    hh, li = lanes // 32, lanes % 32
    for K in (256, 304):
        KS = (K + 63) // 64 * 4
        W = torch.randn(256, K, generator=g) * 0.1
        x = torch.randn(32, K, generator=g).numpy().astype(np.float64)  # the wave's 32 rows
        b = torch.randn(256, generator=g).numpy().astype(np.float64)
        pack = pack_gemm_r6(W).float().numpy().reshape(KS, 8, 3, 2, 32, 8).astype(np.float64)
        A_frag = pack.sum(2).reshape(KS, 8, 64, 8)  # limbs add up exactly; lane = 32 h + i
        acc = np.zeros((8, 16, 64))
        for nt in range(8):
            for gq in range(4):
                for r in range(4):
                    acc[nt, 4 * gq + r] = b[32 * nt + 8 * gq + 4 * hh + r]
        for e in range(KS):
            ec = e if e < K // 16 else 0  # padded k steps: the kernel requests k step 0 again (zero weights)
            B_lane = np.zeros((64, 8))
            for i in range(8):
                B_lane[:, i] = x[li, 16 * ec + 8 * (i // 4) + 4 * hh + i % 4]
            for nt in range(8):
                mfma_32x32x16(A_frag[e, nt], B_lane, acc[nt])
        ref = x @ W.numpy().astype(np.float64).T + b  # [32 points][256]
        for nt in range(8):
            for gq in range(4):
                for r in range(4):
                    got = acc[nt, 4 * gq + r]  # [64 lanes]: point li, output 32 nt + 8 gq + 4 hh + r
                    want = ref[li, 32 * nt + 8 * gq + 4 * hh + r]
                    assert np.abs(got - want).max() < 1e-4 * np.abs(ref).max(), (K, nt, gq, r)


def test_wgrad_r6_lds_image_and_fragment_addresses():
    """csrc/wgrad_r6.hip restated on the CPU: the LDS-DMA placement of a step's raw rows (lane l of the wave that owns row r
    fetches the 16-byte chunk l ^ 8 (r >> 3) and it lands lane-linear) and the fragment reads (base address XOR (f << 7),
    + 1 KiB per point) must hand lane (hh, li) of wave (wn, .) the values M[8 hh + e][128 wn + 32 f + li], e = 0..7 -- and a
    wave-wide read of one e must touch 64 distinct banks."""
    rows = np.arange(16 * 256, dtype=np.float64).reshape(16, 256)  # M[r][c] = 256 r + c
    lds = np.full(16 * 1024 // 4, -1.0)  # one operand's 16 KiB image, in floats
    for half in range(2):  # the two waves that fetch this operand: rows 8 half .. 8 half + 7
        for piece in range(8):
            r = 8 * half + piece
            for l in range(64):
                chunk = l ^ (8 * half)
                dst = (r * 1024 + l * 16) // 4
                lds[dst:dst + 4] = rows[r, 4 * chunk:4 * chunk + 4]
    assert (lds >= 0).all()
    for wn in range(2):
        for f in range(4):
            for e in range(8):
                banks = set()
                for lane in range(64):
                    hh, li = lane // 32, lane % 32
                    rd = (8 * hh) * 1024 + ((128 * wn + li) ^ (32 * hh)) * 4
                    addr = (rd ^ (f << 7)) + e * 1024
                    assert lds[addr // 4] == rows[8 * hh + e, 128 * wn + 32 * f + li], (wn, f, e, lane)
                    banks.add((addr // 4) % 64)
                assert len(banks) == 64, (wn, f, e)


def test_rnarrow_lds_images_fragment_addresses_and_result_layout():
    """csrc/rnarrow.hip restated on the CPU for one 32-point tile and N = 39 outputs (two output tiles): the LDS-DMA placement of
    a K quarter's raw rows (piece i, lane l -> row 4 i + l / 16, physical chunk l % 16 receives the logical chunk
    (l % 16) ^ (row % 16)), the A-fragment reads (row li, logical chunks 4 ks + 2 hh, + 1 at physical chunk ^ (li % 16); the 16
    rows of a read phase on 16 distinct bank groups), the planes of W ([limb][n-tile][k-step][lane]: output 32 nt + l % 32,
    k = 16 ks + 8 (l / 32) ..), the MFMA operand roles (A = points, B = outputs) and the epilogue's element addresses must
    reproduce C = A W^T."""
    rng = np.random.default_rng(5)
    N, NT = 39, 2
    A = rng.standard_normal((32, 256))
    W = rng.standard_normal((N, 256))
    # W planes (limbs not modelled: one exact plane)
    wl = np.zeros((NT, 16, 64, 8))
    for u in range(NT * 16 * 64):
        l, ks, nt = u & 63, (u >> 6) & 15, u >> 10
        n, k0 = 32 * nt + (l & 31), 16 * ks + 8 * (l >> 5)
        if n < N:
            wl[nt, ks, l] = W[n, k0:k0 + 8]
    acc = np.zeros((NT, 16, 64))
    for q in range(4):
        stage = np.full(32 * 64, np.nan)  # [32 rows][64 floats], in floats
        for i in range(8):
            for l in range(64):
                row, pch = 4 * i + (l >> 4), l & 15
                src = 64 * q + 4 * (pch ^ (row & 15))
                dst = (i * 1024 + l * 16) // 4
                assert dst == row * 64 + pch * 4
                stage[dst:dst + 4] = A[row, src:src + 4]
        assert not np.isnan(stage).any()
        for ks in range(4):
            A_lane = np.zeros((64, 8))
            for phase in range(4):  # ds_read_b128: 16 lanes per phase
                groups = set()
                for l in range(16 * phase, 16 * phase + 16):
                    hh, li = l >> 5, l & 31
                    byte = li * 256 + (((4 * ks + 2 * hh) ^ (li & 15)) << 4)
                    groups.add((byte // 16) % 16)
                assert len(groups) == 16, (q, ks, phase)
            for l in range(64):
                hh, li = l >> 5, l & 31
                for c in range(2):
                    byte = li * 256 + (((4 * ks + 2 * hh + c) ^ (li & 15)) << 4)
                    A_lane[l, 4 * c:4 * c + 4] = stage[byte // 4:byte // 4 + 4]
                assert np.array_equal(A_lane[l], A[li, 64 * q + 16 * ks + 8 * hh:64 * q + 16 * ks + 8 * hh + 8])
            for nt in range(NT):
                mfma_32x32x16(A_lane, wl[nt, 4 * q + ks], acc[nt])
    ref = A @ W.T
    for nt in range(NT):
        for l in range(64):
            hh, li = l >> 5, l & 31
            col = 32 * nt + li
            for r in range(16):
                p = 4 * hh + 8 * (r >> 2) + (r & 3)  # the epilogue's o[(8 (r >> 2) + (r & 3)) ldc] from row p0 + 4 hh
                if col < N:
                    assert abs(acc[nt, r, l] - ref[p, col]) < 1e-10, (nt, l, r)
                else:
                    assert acc[nt, r, l] == 0.0
