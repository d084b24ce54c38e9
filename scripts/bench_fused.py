"""time the fused SDF trunk only (env HOLD_FUSED_VARIANT / HOLD_FUSED_STAGGER / HOLD_FUSED_DEBUG are read per process)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hold_amd import field as F, kernels as K, synthetic as syn
dev = "cuda:0"; P = 128 * 16384
sc = syn.make_scene(2)
sd = {k: torch.as_tensor(v).to(dev) for k, v in syn.make_state_dict(sc).items()}
spec = F.FieldSpec("object"); pre = "nodes.object."
eff = lambda p: sd[p + ".weight_v"] * (sd[p + ".weight_g"] / sd[p + ".weight_v"].norm(dim=1, keepdim=True))
iw = [eff(pre + f"implicit_network.lin{l}") for l in range(9)]; ib = [sd[pre + f"implicit_network.lin{l}.bias"] for l in range(9)]
rw = [eff(pre + f"rendering_network.lin{l}") for l in range(5)]; rb = [sd[pre + f"rendering_network.lin{l}.bias"] for l in range(5)]
pk = F.pack_weights(spec, iw, ib, rw, rb, need_bwd=False)
xc = torch.zeros(P, 4, device=dev); xc[:, :3] = torch.rand(P, 3, device=dev) * 1.6 - 0.8
out = torch.empty(P, 1, device=dev)
wpack, bias8 = pk["fused"]
flops = 2.0 * P * (40 * 256 + 6 * 65536 + 217 * 256 + 256)
fn = lambda: K.fused_sdf(xc, P, wpack, bias8, pk["w8_sdf"], float(pk["b8_sdf"]), None, out)
if os.environ.get("HOLD_X6") == "1":  # split-precision, 64-point blocks, pre-split weight limbs
    x6 = F.pack_x6(pk["W"][:8])
    fn = lambda: K.fused_sdf_x6(xc, P, x6, bias8, pk["w8_sdf"], float(pk["b8_sdf"]), None, out)
for _ in range(2): fn()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5): fn()
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 5
print(f"fused P={P}: {ms:.3f} ms {flops / ms / 1e9:.1f} TFLOP/s chk={out.double().sum().item():.6f}", flush=True)
