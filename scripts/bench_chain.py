"""time hold_chain per mode against the equivalent hold_gemm_nt sequences (P = 16384 rays x 98 samples)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hold_amd import kernels as K, gemm as G
dev = "cuda:0"; P = int(sys.argv[1]) if len(sys.argv) > 1 else 16384 * 98
def pack(mats):
    return torch.cat([m.reshape(8, 32, m.shape[1] // 8, 2, 4).permute(2, 0, 3, 1, 4).reshape(-1) for m in mats]).contiguous()
g = torch.Generator().manual_seed(0)
W = [torch.randn(256, 40, generator=g).to(dev) / 6] + [torch.randn(256, 256, generator=g).to(dev) / 16 for _ in range(7)]
b = [torch.randn(256, generator=g).to(dev) * 0.05 for _ in range(8)]
x0 = torch.randn(P, 40, device=dev)
bufs = lambda n: [torch.empty(P, 256, device=dev) for _ in range(n)]
h, t, a2, o1 = bufs(8), bufs(8), bufs(8), bufs(8)
for x in h: x.uniform_(0, 0.05)
for x in t + a2: x.normal_()
wf, wb = pack(W), pack(W[1:])
from hold_amd import field as F
xf, xb = (F.pack_x6(W, 48), F.pack_x6(W[1:], 256)) if os.environ.get("HOLD_X6") == "1" else (None, None)  # hold_chain_x6
def timeit(name, fn, flops):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3): fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 3
    print(f"{name:28s} {ms:8.3f} ms  {flops / ms / 1e9:6.1f} TFLOP/s", flush=True)
f8 = 2.0 * P * 256 * (40 + 7 * 256); f7 = 2.0 * P * 256 * 7 * 256
timeit("chain SOFTPLUS (8 layers)", lambda: K.chain(K.CHAIN_SOFTPLUS, P, x0, wf, 8, 5, skip_layer=3, side=x0, bias=b, out=o1, wpack_x6=xf), f8)
timeit("chain SOFTPLUS no stores", lambda: K.chain(K.CHAIN_SOFTPLUS, P, x0, wf, 8, 5, skip_layer=3, side=x0, bias=b, out=None, wpack_x6=xf), f8)
timeit("chain DSP (7 layers)", lambda: K.chain(K.CHAIN_DSP, P, t[7], wb, 7, 32, skip_layer=3, aux1=h[:7], out=o1[:7], wpack_x6=xb), f7)
timeit("chain DSP+a2 (7 layers)", lambda: K.chain(K.CHAIN_DSP, P, t[7], wb, 7, 32, skip_layer=3, aux1=h[:7], aux2=a2[:7], out=o1[:7], wpack_x6=xb), f7)
timeit("chain DBWD (8 layers)", lambda: K.chain(K.CHAIN_DBWD, P, x0, wf, 8, 5, skip_layer=3, side=x0, aux1=h, aux2=t, out=o1, out2=a2, wpack_x6=xf), f8)
if os.environ.get("HOLD_X6") == "1":
    # register-resident sweeps (hold_chain_r6, csrc/rchain.hip) on the same shapes
    rb = F.pack_r6_stack(torch.stack(W[1:]))
    timeit("r6 DSP (7 layers)", lambda: K.chain(K.CHAIN_DSP, P, t[7], wb, 7, 32, skip_layer=3, aux1=h[:7], out=o1[:7], wpack_x6=xb, wpack_r6=rb), f7)
    timeit("r6 DSP+a2 (7 layers)", lambda: K.chain(K.CHAIN_DSP, P, t[7], wb, 7, 32, skip_layer=3, aux1=h[:7], aux2=a2[:7], out=o1[:7], wpack_x6=xb, wpack_r6=rb), f7)
    rf = F.pack_r6(W[0], torch.stack(W[1:]))
    timeit("r6 DBWD (8 layers)", lambda: K.chain(K.CHAIN_DBWD, P, x0, wf, 8, 5, skip_layer=3, side=x0, aux1=h, aux2=t, out=o1, out2=a2, wpack_x6=xf, wpack_r6=rf), f8)
    # ... and in two fp16 limbs (hold_chain_h3, csrc/rchain_h3.hip; each call includes its conditional f32x6 launch)
    hb, swb = F.pack_h3_stack(torch.stack(W[1:]))
    hf, swf = F.pack_h3(W[0], torch.stack(W[1:]))
    cb, cf = (1.0 / swb).contiguous(), (1.0 / swf).contiguous()
    n0 = K.h3_overflow_count(dev)
    timeit("h3 DSP (7 layers)", lambda: K.chain(K.CHAIN_DSP, P, t[7], wb, 7, 32, skip_layer=3, aux1=h[:7], out=o1[:7], wpack_r6=rb, wpack_h3=hb, c3=cb), f7)
    timeit("h3 DSP+a2 (7 layers)", lambda: K.chain(K.CHAIN_DSP, P, t[7], wb, 7, 32, skip_layer=3, aux1=h[:7], aux2=a2[:7], out=o1[:7], wpack_r6=rb, wpack_h3=hb, c3=cb), f7)
    timeit("h3 DBWD (8 layers)", lambda: K.chain(K.CHAIN_DBWD, P, x0, wf, 8, 5, skip_layer=3, side=x0, aux1=h, aux2=t, out=o1, out2=a2, wpack_r6=rf, wpack_h3=hf, c3=cf), f8)
    print("f32x6 fallbacks during the h3 timings:", K.h3_overflow_count(dev) - n0)
    sys.exit(0)
def layered_sp():
    G.gemm_nt(x0, W[0], o1[0], bias=b[0], epi=G.EPI_SOFTPLUS, K=40)
    for l in range(1, 8): G.gemm_nt(o1[l - 1], W[l], o1[l], bias=b[l], epi=G.EPI_SOFTPLUS)
def layered_dsp():
    cur = t[7]
    for l in range(7): G.gemm_nt(cur, W[l + 1], o1[l], epi=G.EPI_MUL_DSP, aux1=h[l], aux2=a2[l]); cur = o1[l]
timeit("gemm_nt x8 SOFTPLUS", layered_sp, f8)
timeit("gemm_nt x7 DSP+a2", layered_dsp, f7)
