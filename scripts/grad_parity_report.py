"""per-tensor relative error of the HIP training-step gradients against torch autograd on the CPU oracle, in fp32 (what
tests/test_path_gpu.py::test_train_step_gradients_match_oracle_autograd compares against) and in fp64 (the oracle's own
rounding removed).  Prints one line per parameter tensor, worst first."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from parity_common import hip_input, hip_net, ho, oracle_input, setup
from hold_amd import synthetic as syn
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_path_gpu as T

ctx = dict(zip(("sc", "sd_np", "sd", "osc"), setup()))
sc, sd, sdg, osc, b, oinp, rng = T._train_setup(ctx, 6, [0, 2])
gt = torch.from_numpy(b["gt.rgb"]).view(-1, 3)
oo0 = ho.holdnet_forward(osc, sd, {k: (v.detach() if torch.is_tensor(v) else v) for k, v in oinp.items()}, True, rng=rng,
                         current_epoch=25, barf_alpha_iter=4000)
zo = {n: oo0[n + ".z_vals"].detach() for n in sc["entities"]}
grads = {}
for dt in (torch.float32, torch.float64):
    mano = {"right": syn.make_mano_model(True), "left": syn.make_mano_model(False)}
    o = ho.OracleScene(sc, mano, dtype=dt)
    sdd = {k: (v.detach().to(dt).requires_grad_(True) if v.dtype.is_floating_point else v) for k, v in sd.items()}
    _, oi = oracle_input(sc, sdd, [0, 2], 6, 6)
    oi = {k: (v.to(dt) if torch.is_tensor(v) and v.dtype.is_floating_point else v) for k, v in oi.items()}
    r = {k: ({kk: (vv.to(dt) if torch.is_tensor(vv) else vv) for kk, vv in v.items()} if isinstance(v, dict) else v.to(dt)) for k, v in rng.items()}
    out = ho.holdnet_forward(o, sdd, oi, True, rng=r, z_override={n: z.to(dt) for n, z in zo.items()}, current_epoch=25,
                             barf_alpha_iter=4000, stable_merge=True)
    T._loss(out, gt.to(dt)).backward()
    grads[dt] = {k: v.grad.detach().double() for k, v in sdd.items() if torch.is_tensor(v) and v.grad is not None}
if not torch.cuda.is_available():
    print("oracle legs ok:", len(grads[torch.float32]), len(grads[torch.float64])); sys.exit(0)
net = hip_net(sc, ctx["sd_np"], train=True)
rng_c = {k: ({kk: (vv.cuda() if torch.is_tensor(vv) else vv) for kk, vv in v.items()} if isinstance(v, dict) else v.cuda()) for k, v in rng.items()}
out = net(hip_input(b, net, epoch=25, step=10), rng=rng_c, z_override={n: z.cuda() for n, z in zo.items()})
T._loss(out, gt.cuda()).backward()
rows = []
for name, p in net.named_parameters():
    if name in grads[torch.float64] and p.grad is not None:
        g = p.grad.detach().cpu().double()
        r32 = float((g - grads[torch.float32][name]).norm() / (grads[torch.float32][name].norm() + 1e-30))
        r64 = float((g - grads[torch.float64][name]).norm() / (grads[torch.float64][name].norm() + 1e-30))
        o32 = float((grads[torch.float32][name] - grads[torch.float64][name]).norm() / (grads[torch.float64][name].norm() + 1e-30))
        rows.append((r64, r32, o32, name, float(grads[torch.float64][name].norm())))
print("hip-vs-fp64  hip-vs-fp32oracle  fp32oracle-vs-fp64  |grad|  name")
for r64, r32, o32, name, nrm in sorted(rows, reverse=True):
    print(f"{r64:10.2e} {r32:10.2e} {o32:10.2e} {nrm:10.2e}  {name}")
