import os, sys, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hold_amd import field as F, kernels as K, synthetic as syn
dev = "cuda:0"
sc = syn.make_scene(2)
sd = {k: torch.as_tensor(v).to(dev) for k, v in syn.make_state_dict(sc, perturb=0.05).items()}
spec = F.FieldSpec("object"); pre = "nodes.object."
eff = lambda p: sd[p + ".weight_v"] * (sd[p + ".weight_g"] / sd[p + ".weight_v"].norm(dim=1, keepdim=True))
iw = [eff(pre + f"implicit_network.lin{l}") for l in range(9)]; ib = [sd[pre + f"implicit_network.lin{l}.bias"] for l in range(9)]
rw = [eff(pre + f"rendering_network.lin{l}") for l in range(5)]; rb = [sd[pre + f"rendering_network.lin{l}.bias"] for l in range(5)]
pk = F.pack_weights(spec, iw, ib, rw, rb, need_bwd=False)
x6 = F.pack_x6(pk["W"][:8])
wpack, bias8 = pk["fused"]
for P, var, split in ((1000, "0", "rne"), (128 * 4096, "0", "rne"), (1000, "0", "trunc"), (128 * 4096, "0", "trunc"),
                      (1000, "1", "rne"), (128 * 4096, "1", "rne")):
    os.environ["HOLD_FUSED_X6_VARIANT"] = var
    os.environ["HOLD_X6_SPLIT"] = split
    xc = torch.zeros(P, 4, device=dev); xc[:, :3] = torch.rand(P, 3, device=dev) * 1.6 - 0.8
    ref = torch.empty(P, 1, device=dev); out = torch.full((P, 1), 7.0, device=dev)
    try:
        K.fused_sdf(xc, P, wpack, bias8, pk["w8_sdf"], float(pk["b8_sdf"]), None, ref)
        K.fused_sdf_x6(xc, P, x6, bias8, pk["w8_sdf"], float(pk["b8_sdf"]), None, out)
        torch.cuda.synchronize()
        d = (out - ref).abs()
        print("variant", var, "split", split, "P", P, "maxerr", float(d.max()), "mean", float(d.mean()), "refmax", float(ref.abs().max()), "n7", int((out == 7).sum()),
              "nan", int(torch.isnan(out).sum()), "first", out[:4, 0].tolist(), ref[:4, 0].tolist(), flush=True)
        if P > 1000:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3): K.fused_sdf_x6(xc, P, x6, bias8, pk["w8_sdf"], float(pk["b8_sdf"]), None, out)
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 3
            print("x6 ms", ms, "TF-eq", 2.0 * P * (40 * 256 + 6 * 65536 + 217 * 256 + 256) / ms / 1e9, flush=True)
    except Exception as ex:
        print("EXC", type(ex).__name__, str(ex)[:300], flush=True)
