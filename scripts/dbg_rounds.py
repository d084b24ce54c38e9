"""histogram of sampler rounds per 1280-ray group over a 512x512 frame (training-mode sampling)"""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch, numpy as np
from parity_common import *
from hold_amd import synthetic as syn
import hold_amd
sc = syn.make_scene(8); sd_np = syn.make_state_dict(sc, barf_iter=3999)
net = hold_amd.build_from_scene(sc, sd_np, device="cuda:0")
for node in net.nodes.values():
    node.implicit_network.embedder_obj.step(); node.ray_sampler.rng_device = "cuda"
net.train()
G = int(sys.argv[1]) if len(sys.argv) > 1 else 1280
uv = syn.make_uv(512, 512)
b = syn.make_batch(sc, [0], uv, 512, 512)
inp = {k: torch.from_numpy(v).cuda() for k, v in b.items()}
from hold_amd.train import chunked_input
hist = {n: collections.Counter() for n in net.nodes}
with torch.no_grad():
    for lo in range(0, 262144, G):
        c = chunked_input(inp, lo, lo + G); c["current_epoch"] = 0; c["global_step"] = 0
        for node in net.nodes.values(): c.update(node.params(c["idx"]))
        net(c)
        for n, node in net.nodes.items(): hist[n][node.ray_sampler.last_iters] += 1
for n, h in hist.items():
    tot = sum(h.values()); avg = sum(k * v for k, v in h.items()) / tot
    print(n, dict(sorted(h.items())), "avg rounds %.2f" % avg)
