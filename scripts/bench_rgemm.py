"""time hold_gemm_r6 (csrc/rgemm.hip) against hold_gemm_nt_x6 on the rendering net's shapes (P = 16384 rays x 98 samples)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hold_amd import field as F, gemm as G
dev = "cuda:0"; P = int(sys.argv[1]) if len(sys.argv) > 1 else 16384 * 98
def timeit(fn, n=5):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
g = torch.Generator().manual_seed(0)
for K, epi in ((256, "relu"), (256, "mask"), (256, "none"), (304, "relu")):
    A = torch.randn(P, K, device=dev); W = (torch.randn(256, K, generator=g) / 16).to(dev); b = torch.randn(256, device=dev)
    aux = torch.randn(P, 256, device=dev) if epi == "mask" else None
    o1 = torch.empty(P, 256, device=dev); o2 = torch.empty(P, 256, device=dev)
    pk = F.pack_gemm_r6(W)
    e_nt = {"relu": G.EPI_RELU, "mask": G.EPI_MUL_DRELU, "none": G.EPI_NONE}[epi]
    e_r6 = {"relu": G.R6_RELU, "mask": G.R6_MASK, "none": G.R6_NONE}[epi]
    t1 = timeit(lambda: G.gemm_nt(A, W, o1, bias=None if epi == "mask" else b, epi=e_nt, aux1=aux, K=K))
    t2 = timeit(lambda: G.gemm_r6(A, pk, o2, K=K, bias=None if epi == "mask" else b, epi=e_r6, aux=aux))
    fl = 2.0 * P * 256 * K
    pk3, c3 = F.pack_gemm_h3(W)
    am_in, am_out, o3 = A.abs().amax(1).contiguous(), torch.empty(P, device=dev), torch.empty(P, 256, device=dev)
    t3 = timeit(lambda: G.gemm_h3(A, pk3, c3, o3, K=K, wpack_r6=pk, bias=None if epi == "mask" else b, epi=e_r6, aux=aux, amax_in=am_in, amax_out=am_out))
    print(f"K={K} {epi:5s} P={P}: gemm_h3 {t3:.3f} ms {2.0 * P * 256 * K / t3 / 1e9:.1f} TF-eq ({t2 / t3:.2f} x gemm_r6) | max diff to r6 {float((o3 - o2).abs().max()):.2e}", flush=True)
    print(f"K={K} {epi:5s} P={P}: gemm_nt_x6 {t1:.3f} ms {fl / t1 / 1e9:.1f} TF-eq | gemm_r6 {t2:.3f} ms {fl / t2 / 1e9:.1f} TF-eq | max diff {float((o1 - o2).abs().max()):.2e}", flush=True)
