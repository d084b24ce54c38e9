"""stage-by-stage HIP vs oracle deviations (run on the GPU box)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from parity_common import *

W = H = int(sys.argv[1]) if len(sys.argv) > 1 else 8
sc, sd_np, sd, osc = setup()
b, oinp = oracle_input(sc, sd, [1, 3], W, H)
ex = {}
oo = ho.holdnet_forward(osc, sd, oinp, False, extras=ex, stable_merge=True)
net = hip_net(sc, sd_np)
zo = {n: oo[n + ".z_vals"].cuda() for n in sc["entities"]}
out = net(hip_input(b, net), z_override=zo)
fac = net._last_factors
for n in sc["entities"]:
    f, e = fac[n], ex[n]
    print(f"[{n}] x_c {rel_err(f['canonical_pts'], e['x_c']):.2e} sdf {rel_err(f['sdf'].view(-1,1), e['sdf']):.2e} "
          f"grad {rel_err(net.nodes[n].field.saved['g'][:, :3], e['grad']):.2e} "
          f"feat {rel_err(net.nodes[n].field.saved['rin'][:, 14:270], e['feat']):.2e}")
    if 'tfs' in e: print(f"     tfs {rel_err(f['tfs'], e['tfs']):.2e}")
for k in ["rgb", "fg_rgb", "normal", "depth", "mask_prob", "bg_rgb_only", "semantics", "right.fg_rgb", "object.fg_rgb",
          "right.normal", "object.normal", "fg_weights", "bg_weights", "right.bg_weights", "object.depth"]:
    print(f"{k:16s} max abs {float((out[k].detach().cpu() - oo[k].detach()).abs().max()):.3e}")
# full sampler
out2 = net(hip_input(b, net))
for n in sc["entities"]:
    dz = (out2[n + ".z_vals"].cpu() - oo[n + ".z_vals"]).abs()
    print(f"[{n}] sampler: iters hip={net.nodes[n].ray_sampler.last_iters} oracle={ex[n]['iters']} "
          f"max dz {float(dz.max()):.3e} frac(dz>1e-3) {float((dz > 1e-3).float().mean()):.4f}")
mse = float(((out2["rgb"].cpu() - oo["rgb"]) ** 2).mean())
print("full-pipeline PSNR vs oracle: %.2f dB" % (10 * np.log10(1.0 / max(mse, 1e-20))))
