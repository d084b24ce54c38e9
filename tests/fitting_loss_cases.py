"""shared by the CPU (oracle vs golden) and GPU (hold_amd.fitting vs golden) checks of the pose-refinement losses;
tests/golden/fitting_losses.npz holds inputs and the REFERENCE's own outputs / gradients
(scripts/make_golden_fitting.py, code/src/fitting/loss.py:84-165)."""
import numpy as np
import torch


def run_two_hand(g, loss_fn_ih, dev):
    """two successive calls (the second one sees the cached 2-D targets and moved hands) -> max relative deviations"""
    t = lambda a: torch.as_tensor(np.asarray(a)).to(dev)
    cidx = t(g["contact_idx"]).long()
    tg = {k: t(g["tg." + k]) for k in ("right", "left", "object")}
    worst = 0.0
    for it in range(2):
        out = {"K": t(g["in.K"]), "object.v3d_c": t(g[f"ih{it}.in_obj"]).requires_grad_(True),
               "object.mask": t(g["in.object.mask"]).requires_grad_(True),
               "right.v3d_c": t(g[f"ih{it}.in"][0]).requires_grad_(True),
               "left.v3d_c": t(g[f"ih{it}.in"][1]).requires_grad_(True)}
        d = loss_fn_ih(out, tg, cidx)
        for k in ("mask_o", "v2d_r", "v2d_l", "contact_ro", "contact_lo", "loss"):
            ref = float(g[f"ih{it}.{k}"])
            worst = max(worst, abs(float(d[k]) - ref) / max(1.0, abs(ref)))
        d["loss"].backward()
        for k in ("right.v3d_c", "left.v3d_c", "object.v3d_c", "object.mask"):
            rg = g[f"ih{it}.grad.{k}"]
            worst = max(worst, float((out[k].grad.cpu() - torch.as_tensor(rg)).abs().max()) / max(1.0, float(np.abs(rg).max())))
    return worst


def run_single_hand(g, loss_fn_h, dev):
    t = lambda a: torch.as_tensor(np.asarray(a)).to(dev)
    cidx = t(g["contact_idx"]).long()
    tg = {k: t(g["tg." + k]) for k in ("right", "object")}
    out = {k: t(g["in." + k]).requires_grad_(True) for k in ("right.v3d_c", "object.v3d_c", "object.mask", "right.mask")}
    d = loss_fn_h(out, tg, "right", cidx)
    worst = 0.0
    for k in ("mask_o", "mask_h", "fine_ho", "loss"):
        worst = max(worst, abs(float(d[k]) - float(g["rh." + k])) / max(1.0, abs(float(g["rh." + k]))))
    d["loss"].backward()
    for k in out:
        rg = g["rh.grad." + k]
        worst = max(worst, float((out[k].grad.cpu() - torch.as_tensor(rg)).abs().max()) / max(1.0, float(np.abs(rg).max())))
    return worst
