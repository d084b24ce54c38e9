"""time the register-resident trunk (csrc/rmlp.hip) against the LDS-resident kernels it replaces: the sampler's SDF query
(hold_fused_sdf_x6 vs hold_fused_sdf_r6) and the training forward trunk (embed + hold_chain_x6(SOFTPLUS) vs hold_trunk_r6)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hold_amd import field as F, kernels as K, synthetic as syn
dev = "cuda:0"
sc = syn.make_scene(2)
sd = {k: torch.as_tensor(v).to(dev) for k, v in syn.make_state_dict(sc).items()}
spec = F.FieldSpec("object"); pre = "nodes.object."
eff = lambda p: sd[p + ".weight_v"] * (sd[p + ".weight_g"] / sd[p + ".weight_v"].norm(dim=1, keepdim=True))
iw = [eff(pre + f"implicit_network.lin{l}") for l in range(9)]; ib = [sd[pre + f"implicit_network.lin{l}.bias"] for l in range(9)]
pk = F.pack_weights(spec, iw, ib, None, None, need_bwd=False)
bias8 = pk["fused"][1]


def timeit(fn, n=5):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


PS = [int(a) for a in sys.argv[1:]] or [128 * 16384, 128 * 1280, 98 * 16384]
for P in PS:
    xc = torch.zeros(P, 4, device=dev); xc[:, :3] = torch.rand(P, 3, device=dev) * 1.6 - 0.8
    out = torch.empty(P, 1, device=dev); out2 = torch.empty(P, 1, device=dev)
    flops = 2.0 * P * (40 * 256 + 6 * 65536 + 217 * 256 + 256)
    b8f = float(pk["b8_sdf"])
    ms = timeit(lambda: K.fused_sdf_x6(xc, P, pk["fused_x6"], bias8, pk["w8_sdf"], b8f, None, out))
    ms2 = timeit(lambda: K.fused_sdf_r6(xc, P, pk["trunk_r6"], bias8, pk["w8_sdf"], pk["b8_sdf"], None, out2))
    pk3, sw = F.pack_h3(pk["W"][0], torch.stack([torch.nn.functional.pad(w, (0, 0, 0, 256 - w.shape[0])) for w in pk["W"][1:8]]))
    bs3, c33 = (bias8 * (sw * F.H3_ACT_SCALE).view(8, 1)).contiguous(), (1.0 / sw).contiguous()
    out3 = torch.empty(P, 1, device=dev)
    ms3 = timeit(lambda: K.fused_sdf_h3(xc, P, pk3, bs3, c33, pk["w8_sdf"], pk["b8_sdf"], None, out3))
    print(f"sdf query P={P}: h3 {ms3:.3f} ms {flops / ms3 / 1e9:.1f} TF-eq ({ms2 / ms3:.2f}x r6) | max diff to r6 {float((out3 - out2).abs().max()):.2e}", flush=True)
    print(f"sdf query P={P}: x6p {ms:.3f} ms {flops / ms / 1e9:.1f} TF-eq | r6 {ms2:.3f} ms {flops / ms2 / 1e9:.1f} TF-eq | "
          f"max diff {float((out - out2).abs().max()):.2e}", flush=True)
    if P * 256 * 4 * 9 < 60e9:
        h = [torch.empty(P, 256, device=dev) for _ in range(8)]
        h2 = [torch.empty(P, 256, device=dev) for _ in range(8)]
        in0 = torch.empty(P, 40, device=dev)
        def chain():
            K.embed_fwd(xc, 3, 6, P, in0, out2=h[3][:, 217:])
            K.chain(K.CHAIN_SOFTPLUS, P, in0, pk["fused"][0], 8, 5, skip_layer=3, side=in0, bias=[bias8[l] for l in range(8)],
                    out=h, wpack_x6=pk["chain_fwd_x6"])
        ms = timeit(chain)
        ms2 = timeit(lambda: K.trunk_r6(xc, P, pk["trunk_r6"], bias8, None, h2))
        fl = 2.0 * P * (40 * 256 + 6 * 65536 + 217 * 256)
        h3b = [torch.empty(P, 256, device=dev) for _ in range(8)]
        ms3 = timeit(lambda: K.trunk_h3(xc, P, pk3, bs3, c33, None, h3b))
        print(f"fwd trunk P={P}: trunk_h3 {ms3:.3f} ms {fl / ms3 / 1e9:.1f} TF-eq ({ms2 / ms3:.2f}x trunk_r6) | max diff to r6 "
              f"{max(float((a - b).abs().max()) for a, b in zip(h3b, h2)):.2e}", flush=True)
        del h3b
        dl = [float((a - b).abs().max()) for a, b in zip(h, h2)]
        d = max(dl)
        if d > 1e-3:  # diagnostics: which layer / column, and which of the two kernels is off a torch fp32 trunk on 512 rows
            l = dl.index(d); idx = int((h[l] - h2[l]).abs().argmax()); r, c = idx // 256, idx % 256
            print(f"  per-layer max diff {['%.1e' % v for v in dl]}; layer {l} row {r} col {c}: chain {float(h[l][r, c]):.6f} r6 {float(h2[l][r, c]):.6f}")
            rows = slice(r - r % 128, r - r % 128 + 128)
            cur = in0[rows, :39].double(); emb = cur
            for ll in range(8):
                W = pk["W"][ll].double()
                cur = torch.nn.functional.softplus(cur @ W.t()[:cur.shape[1]] + pk["b"][ll].double()[:W.shape[0]], beta=100)
                if ll == 3: cur = torch.cat([cur[:, :217], emb], 1)
                print(f"    layer {ll}: |chain - ref| {float((h[ll][rows].double() - cur).abs().max()):.2e}  |r6 - ref| {float((h2[ll][rows].double() - cur).abs().max()):.2e}"
                      f"  worst r6 col {int((h2[ll][rows].double() - cur).abs().max(0).values.argmax())}")
        print(f"fwd trunk P={P}: embed+chain_x6 {ms:.3f} ms {fl / ms / 1e9:.1f} TF-eq | trunk_r6 {ms2:.3f} ms {fl / ms2 / 1e9:.1f} TF-eq | "
              f"max diff {d:.2e}", flush=True)
