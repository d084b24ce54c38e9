"""stress probe for the hand-counted vector-memory waits of hold_trunk_h3 (csrc/rmlp_h3.hip, STORE variant): the training trunk on
1.6 M points repeated N times with other kernels in between; every launch must reproduce the first one's eight h matrices BIT FOR BIT
(a ring slot read before its pieces landed shows up as a few differing rows, as it did for hold_gemm_h3 in GPU call 16)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import torch
from hold_amd import field as F, kernels as K
from test_rmlp_gpu import _net, _h3_pack
dev = torch.device("cuda:0")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
P = int(sys.argv[2]) if len(sys.argv) > 2 else 16384 * 98
w0, S, bias, w8, bw = _net(11, dev, True)
pk, bs, c3 = _h3_pack(w0, S, bias)
g = torch.Generator().manual_seed(3)
xc = torch.zeros(P, 4)
xc[:, :3] = torch.rand(P, 3, generator=g) * 1.6 - 0.8
xc = xc.to(dev)
ref = [torch.empty(P, 256, device=dev) for _ in range(8)]
h = [torch.empty(P, 256, device=dev) for _ in range(8)]
K.trunk_h3(xc, P, pk, bs, c3, bw, ref)
torch.cuda.synchronize()
junk = torch.randn(4096, 4096, device=dev)
bad = 0
for it in range(N):
    if it % 3 == 1:
        junk = (junk * 1.0001).contiguous()  # some other traffic through the L2 in between
    K.trunk_h3(xc, P, pk, bs, c3, bw, h)
    diff = [int((a != b).any(1).sum()) for a, b in zip(h, ref)]
    if any(diff):
        bad += 1
        print(f"iteration {it}: differing rows per layer {diff}", flush=True)
print(f"mismatching iterations: {bad} of {N} guard {K.h3_guard(dev).tolist()}")
