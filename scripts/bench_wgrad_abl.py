"""developer-build timing ablations of the whole-dW weight gradient (csrc/wgrad_r6.hip): HOLD_WGRAD_ABL = 0 (the kernel),
1 (no fragment reads / limb splits), 2 (no LDS-DMA), 3 (no MFMAs).  Needs HOLD_LIB=<libholdhip_dev.so>.  Results of the
ablated kernels are garbage; only their durations mean something."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import hold_amd
from hold_amd import gemm

dev = torch.device("cuda:0")
P = int(sys.argv[1]) if len(sys.argv) > 1 else 1638400
R = torch.randn(P, 256, device=dev)
X = torch.randn(P, 256, device=dev)
dW = torch.empty(256, 256, device=dev)
for _ in range(3):
    gemm.wgrad(R, X, dW, None)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    gemm.wgrad(R, X, dW, None)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 20
print(f"HOLD_WGRAD_ABL={os.environ.get('HOLD_WGRAD_ABL', '0')} wgrad P={P} 256x256 (launch + reduction): {ms:.3f} ms  "
      f"{2.0 * P * 65536 / ms / 1e9:.1f} TF-eq  {P * 2048 / ms / 1e9:.2f} TB/s of operand rows")
