import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from parity_common import *
from hold_amd.train import train_step
sc, sd_np, sd, osc = setup()
net = hip_net(sc, sd_np, train=True)
for node in net.nodes.values(): node.ray_sampler.rng_device = "cuda"
res = int(sys.argv[1]); chunk = int(sys.argv[2])
uv = syn.make_uv(res, res)
b = syn.make_batch(sc, [1], uv, res, res)
inp = {k: torch.from_numpy(v).cuda() for k, v in b.items()}
for it in range(2):
    t0 = time.time()
    loss, n = train_step(net, inp, chunk, step=it)
    torch.cuda.synchronize()
    print("step", it, "loss", loss, "rays", n, "time %.3f" % (time.time() - t0), flush=True)
