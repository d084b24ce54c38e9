"""same-box A/B of the whole-dW weight gradient: hold_wgrad_x6 (wgrad_r6_kernel, three bf16 limbs / six products) against
hold_wgrad_h3 (wgrad_h3_kernel, two fp16 limbs / three products, per-workgroup scales) on N(0,1) operands and on operands
shaped like the step's (softplus outputs x small cotangents), with the error of both against fp64"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import hold_amd
from hold_amd import gemm
dev = "cuda:0"


def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for P in (16384 * 98, 125440):
    for name, mk in (("N(0,1) x N(0,1)", lambda: (torch.randn(P, 256, device=dev), torch.randn(P, 256, device=dev))),
                     ("cotangent 1e-6 x softplus", lambda: (torch.randn(P, 256, device=dev) * 1e-6 * torch.rand(P, 1, device=dev),
                                                            torch.nn.functional.softplus(torch.randn(P, 256, device=dev) * 0.3 - 0.2, beta=100)))):
        R, X = mk()
        ref = None
        if P <= 200000 or name.startswith("N"):
            ref = torch.zeros(256, 256, dtype=torch.float64, device=dev)
            for i in range(0, P, 65536):
                ref += R[i:i + 65536].double().t() @ X[i:i + 65536].double()
        res = {}
        for mode in ("f32x6", "f16x3"):
            hold_amd.set_precision(mode)
            dW = torch.zeros(256, 256, device=dev); db = torch.zeros(256, device=dev)
            ms = timeit(lambda: gemm.wgrad(R, X, dW, db))
            err = float((dW.double() - ref).abs().max() / ref.abs().max()) if ref is not None else float("nan")
            res[mode] = (ms, err)
        a, b = res["f32x6"], res["f16x3"]
        print(f"P={P} {name}: x6 {a[0]:.3f} ms {2.0 * P * 65536 / a[0] / 1e9:.1f} TF-eq err {a[1]:.2e} | h3 {b[0]:.3f} ms "
              f"{2.0 * P * 65536 / b[0] / 1e9:.1f} TF-eq err {b[1]:.2e} | {a[0] / b[0]:.2f}x", flush=True)
