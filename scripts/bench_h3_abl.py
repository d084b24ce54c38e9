"""developer-build timing ablation of hold_fused_sdf_h3 (HOLD_LIB=hold_amd/libholdhip_dev.so, HOLD_H3_ABL=0..3: bit 0 = v_exp_f32
replaced by v_mul_f32, bit 1 = v_log_f32): what the 16 transcendentals of a k step cost beyond 16 ordinary VALU instructions"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hold_amd import field as F, kernels as K
dev = "cuda:0"
g = torch.Generator().manual_seed(0)
W = [torch.randn(256, 40, generator=g).to(dev) / 6] + [torch.randn(256, 256, generator=g).to(dev) / 16 for _ in range(7)]
bias = torch.randn(8, 256, generator=g).to(dev) * 0.05
w8 = (torch.randn(256, generator=g) / 16).to(dev); b8 = torch.full((1,), 0.25, device=dev)
P = 128 * 16384
xc = torch.zeros(P, 4, device=dev); xc[:, :3] = torch.rand(P, 3, device=dev) * 1.6 - 0.8
out = torch.empty(P, 1, device=dev)
pk3, sw = F.pack_h3(W[0], torch.stack(W[1:]))
bs, c3 = (bias * (sw * F.H3_ACT_SCALE).view(8, 1)).contiguous(), (1.0 / sw).contiguous()
fn = lambda: K.fused_sdf_h3(xc, P, pk3, bs, c3, w8, b8, None, out)
for _ in range(3): fn()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): fn()
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 20
print(f"HOLD_H3_ABL={os.environ.get('HOLD_H3_ABL', '0')}: {ms:.3f} ms per launch, {ms * 1e-3 / (P / 128 / 256 * 116) * 1e9:.0f} ns per k step")
