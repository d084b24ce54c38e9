"""micro-benchmark of hold_gemm_nt / hold_wgrad (TFLOP/s on the fp32 matrix cores)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import hold_amd
from hold_amd import gemm
if os.environ.get("HOLD_X6") == "0":
    hold_amd.set_precision("f32")
print("precision", hold_amd.precision())

dev = torch.device("cuda:0")
P = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
for (N, K, epi, name) in [(256, 256, gemm.EPI_NONE, "none"), (256, 256, gemm.EPI_SOFTPLUS, "softplus"),
                          (256, 256, gemm.EPI_MUL_DSP, "mul_dsp"), (256, 40, gemm.EPI_SOFTPLUS, "k40"),
                          (3, 256, gemm.EPI_SIGMOID, "n3")]:
    A = torch.randn(P, K, device=dev)
    W = torch.randn(N, K, device=dev) / 16
    b = torch.randn(N, device=dev)
    out = torch.empty(P, N, device=dev)
    aux = torch.rand(P, N, device=dev) * 0.05 if epi == gemm.EPI_MUL_DSP else None
    for _ in range(3):
        gemm.gemm_nt(A, W, out, bias=b, epi=epi, aux1=aux)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        gemm.gemm_nt(A, W, out, bias=b, epi=epi, aux1=aux)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print(f"gemm_nt {name:9s} P={P} N={N} K={K}: {ms:.3f} ms  {2.0 * P * N * K / ms / 1e9:.1f} TFLOP/s  "
          f"{(P * (K + N) * 4) / ms / 1e6:.0f} GB/s")
R = torch.randn(P, 256, device=dev)
X = torch.randn(P, 256, device=dev)
dW = torch.empty(256, 256, device=dev)
db = torch.empty(256, device=dev)
for _ in range(3):
    gemm.wgrad(R, X, dW, db)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    gemm.wgrad(R, X, dW, db)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 10
print(f"wgrad P={P} 256x256: {ms:.3f} ms {2.0 * P * 65536 / ms / 1e9:.1f} TFLOP/s")

if not os.environ.get("HOLD_LIB"):
    sys.exit(0)  # the pure-MFMA loop below is a developer-build diagnostics kernel
from hold_amd import _lib
blocks, iters = 512 * 8, 512
o = torch.empty(blocks * 256, device=dev)
for rnd in (0, 1):
    for _ in range(2):
        _lib.call("hold_diag_mfma_peak", _lib.ptr(o), blocks, iters, rnd)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        _lib.call("hold_diag_mfma_peak", _lib.ptr(o), blocks, iters, rnd)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 3
    fl = blocks * 4 * iters * 64 * (2.0 * 32 * 32 * 2)
    print(f"pure MFMA f32 32x32x2 loop (2 waves/SIMD, {'random' if rnd else 'constant'} operands): {ms:.3f} ms {fl / ms / 1e9:.1f} TFLOP/s")

wsrc = torch.randn(65536, device=dev) * 0.05
o2 = torch.empty(256 * 512, device=dev)
for mode in (2, 3):
    it2 = 32 * 8 * 8
    for _ in range(2):
        _lib.call("hold_diag_mfma_lds", _lib.ptr(o2), _lib.ptr(wsrc), 256, it2, mode)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    _lib.call("hold_diag_mfma_lds", _lib.ptr(o2), _lib.ptr(wsrc), 256, it2, mode)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    fl = 256 * 8 * it2 * 16 * (2.0 * 32 * 32 * 2)
    print(f"MFMA fed from LDS (mode {mode}; 8 waves/CU): {ms:.3f} ms {fl / ms / 1e9:.1f} TFLOP/s")

# fused SDF trunk vs layered trunk
from hold_amd import field as F, kernels as K, synthetic as syn
sc = syn.make_scene(2)
sd = {k: torch.as_tensor(v).to(dev) for k, v in syn.make_state_dict(sc).items()}
spec = F.FieldSpec("object"); pre = "nodes.object."
eff = lambda p: sd[p + ".weight_v"] * (sd[p + ".weight_g"] / sd[p + ".weight_v"].norm(dim=1, keepdim=True))
iw = [eff(pre + f"implicit_network.lin{l}") for l in range(9)]; ib = [sd[pre + f"implicit_network.lin{l}.bias"] for l in range(9)]
rw = [eff(pre + f"rendering_network.lin{l}") for l in range(5)]; rb = [sd[pre + f"rendering_network.lin{l}.bias"] for l in range(5)]
pk = F.pack_weights(spec, iw, ib, rw, rb, need_bwd=False)
nf = F.NodeField(spec, dev)
xc = torch.zeros(P, 4, device=dev); xc[:, :3] = torch.rand(P, 3, device=dev) * 1.6 - 0.8
out = torch.empty(P, 1, device=dev)
wpack, bias8 = pk["fused"]
flops = 2.0 * P * (40 * 256 + 2 * 65536 + 217 * 256 + 4 * 65536 + 256)
for name, fn in [("fused", lambda: K.fused_sdf(xc, P, wpack, bias8, pk["w8_sdf"], float(pk["b8_sdf"]), None, out)),
                 ("layered", lambda: (nf._trunk(pk, xc, P, None, False), K.rowdot(nf.pool.get("h_pp1", P, 256), pk["w8_sdf"], 256, 0.0, P, out)))]:
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    print(f"sdf trunk {name:8s} P={P}: {ms:.3f} ms  {flops / ms / 1e9:.1f} TFLOP/s (algorithmic)")
