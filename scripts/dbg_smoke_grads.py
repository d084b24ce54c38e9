"""is the training step of __graft_entry__.smoke() reproducible?  Runs it twice in one process (same seeds) and once per
A/B route, printing the per-parameter gradient norms that differ."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from parity_common import hip_input, hip_net, ho, oracle_input, setup


def grads(seed=0):
    sc, sd_np, sd, osc = setup()
    b, oinp = oracle_input(sc, sd, [1], 6, 6)
    net = hip_net(sc, sd_np)
    net.train()
    torch.manual_seed(seed)
    out = net(hip_input(b, net, epoch=25, step=1))
    loss = (out["rgb"] - torch.from_numpy(b["gt.rgb"]).view(-1, 3).cuda()).abs().mean()
    loss.backward()
    return float(loss), {n: p.grad.detach().clone() for n, p in net.named_parameters() if p.grad is not None}


l0, g0 = grads()
l1, g1 = grads()
print("loss", l0, l1, "sum of norms", sum(float(v.norm()) for v in g0.values()), sum(float(v.norm()) for v in g1.values()))
worst = sorted(((float((g0[n] - g1[n]).abs().max()) / (float(g0[n].abs().max()) + 1e-30), n) for n in g0), reverse=True)[:8]
print("run-to-run, same seed: worst relative differences", worst)
l2, g2 = grads(seed=1)
print("other seed: loss", l2, "sum of norms", sum(float(v.norm()) for v in g2.values()))
top = sorted(((float(v.norm()), n) for n, v in g0.items()), reverse=True)[:6]
print("largest norms", top)
