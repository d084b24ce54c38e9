"""micro-benchmark of hold_gemm_nt / hold_wgrad (TFLOP/s on the fp32 matrix cores)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hold_amd import gemm

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

from hold_amd import _lib
blocks, iters = 512 * 8, 512
o = torch.empty(blocks * 256, device=dev)
for _ in range(2):
    _lib.call("hold_diag_mfma_peak", _lib.ptr(o), blocks, iters)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
_lib.call("hold_diag_mfma_peak", _lib.ptr(o), blocks, iters)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1)
fl = blocks * 4 * iters * 64 * (2.0 * 32 * 32 * 2)
print(f"pure MFMA f32 32x32x2 loop (2 waves/SIMD): {ms:.3f} ms {fl / ms / 1e9:.1f} TFLOP/s")
