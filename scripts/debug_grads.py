"""HIP vs oracle (CPU autograd) gradients of a fwd+bwd training step on identical z_vals / rng draws."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from parity_common import *

W = H = int(sys.argv[1]) if len(sys.argv) > 1 else 6
sc, sd_np, sd, osc = setup()
sdg = {k: (v.clone().requires_grad_(True) if v.dtype.is_floating_point else v) for k, v in sd.items()}
b, oinp = oracle_input(sc, sdg, [0, 2], W, H)
N = 2 * W * H
g = torch.Generator().manual_seed(5)
rng = {"bg_t": torch.rand(N, 32, generator=g)}
for n in sc["entities"]:
    rng[n] = {"t_uniform": torch.rand(N, 128, generator=g), "u_final": torch.rand(N, 64, generator=g),
              "perm": (lambda S, _s=len(rng): torch.randperm(S, generator=torch.Generator().manual_seed(100 + _s)))}
# pass 1: oracle sampler -> z
oo0 = ho.holdnet_forward(osc, sd, {k: (v.detach() if torch.is_tensor(v) else v) for k, v in oinp.items()}, True,
                         rng=rng, current_epoch=25, barf_alpha_iter=4000)
zo = {n: oo0[n + ".z_vals"].detach() for n in sc["entities"]}
oo = ho.holdnet_forward(osc, sdg, oinp, True, rng=rng, z_override=zo, current_epoch=25, barf_alpha_iter=4000,
                        stable_merge=True)
gt = torch.from_numpy(b["gt.rgb"]).view(-1, 3)
def loss_fn(o, gt):
    return ((o["rgb"] - gt).abs().mean() + 0.1 * (o["semantics"] ** 2).mean() + 0.05 * o["normal"].sum(-1).mean()
            + 0.02 * o["right.fg_rgb"].sum(-1).mean() + 0.03 * o["object.mask_prob"].mean() + 0.01 * o["depth"].mean())
lo = loss_fn(oo, gt)
oo["bg_rgb_only"].retain_grad()
lo.backward()
if not torch.cuda.is_available():
    print("oracle part OK, loss", float(lo)); sys.exit(0)
net = hip_net(sc, sd_np, train=True)
rng_c = {k: ({kk: vv.cuda() if kk != "perm" else vv for kk, vv in v.items()} if isinstance(v, dict) else v.cuda())
         for k, v in rng.items()}
out = net(hip_input(b, net, epoch=25, step=10), rng=rng_c, z_override={n: z.cuda() for n, z in zo.items()})
lh = loss_fn(out, gt.cuda())
bgo = out["rgb"] - out["fg_rgb"]
lh.backward()
for k in ["rgb", "fg_rgb", "bg_weights", "bg_z_vals", "semantics", "normal"]:
    print(f"   {k:12s} max abs {float((out[k].detach().cpu() - oo[k].detach()).abs().max()):.3e}")
bg_h = (out["rgb"] - out["fg_rgb"]).detach().cpu() ; bg_o = (oo["rgb"] - oo["fg_rgb"]).detach()
print("   bgw*bg_rgb max abs %.3e" % float((bg_h - bg_o).abs().max()))
print("   oracle d bg_only norm %.4e" % float(oo["bg_rgb_only"].grad.norm()))
print("loss oracle %.7f hip %.7f ; rgb max abs %.3e" % (float(lo), float(lh), float((out["rgb"].cpu() - oo["rgb"]).abs().max())))
worst = 0
for name, p in net.named_parameters():
    if name not in sdg or sdg[name].grad is None:
        if p.grad is not None and p.requires_grad and "human_layer" not in name: print("  (no oracle grad)", name)
        continue
    og = sdg[name].grad
    if p.grad is None:
        print("  MISSING hip grad", name, float(og.norm())); continue
    rel = float((p.grad.cpu() - og).norm() / (og.norm() + 1e-20))
    worst = max(worst, rel)
    flag = "" if rel < 2e-3 else "   <<<<"
    print(f"  {name:58s} rel {rel:.2e} |g| {float(og.norm()):.2e}{flag}")
print("worst", worst)
print("loss oracle %.7f hip %.7f" % (float(lo), float(lh)))
