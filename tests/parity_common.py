"""shared helpers for the GPU parity tests: oracle scene (CPU) + HIP net (cuda:0) on identical inputs."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from hold_amd import synthetic as syn  # noqa: E402
from oracle import hold_oracle as ho  # noqa: E402


def setup(n_frames=4, two_hands=False, barf_iter=3999):
    sc = syn.make_scene(n_frames=n_frames, two_hands=two_hands)
    sd_np = syn.make_state_dict(sc, barf_iter=barf_iter)
    mano = {"right": syn.make_mano_model(True), "left": syn.make_mano_model(False)}
    osc = ho.OracleScene(sc, mano)
    sd = {k: torch.as_tensor(v) for k, v in sd_np.items()}
    return sc, sd_np, sd, osc


def hip_net(sc, sd_np, train=False):
    import hold_amd

    net = hold_amd.build_from_scene(sc, sd_np, device="cuda:0")
    for node in net.nodes.values():
        node.params.defrost()
        node.implicit_network.embedder_obj.step()  # refresh BARF weights from the loaded counter
    if train:
        net.train()
    else:
        net.eval()
        for node in net.nodes.values():
            node.implicit_network.embedder_obj.eval()
    return net


def oracle_input(sc, sd, frames, W, H, requires_grad=False):
    uv = syn.make_uv(W, H)
    b = syn.make_batch(sc, frames, uv, W, H)
    inp = {k: torch.from_numpy(v) for k, v in b.items()}
    idx = inp["idx"]
    for nid in sc["entities"]:
        pre = f"nodes.{nid}.params."
        if nid == "object":
            inp["object.global_orient"] = sd[pre + "global_orient.weight"][idx]
            inp["object.transl"] = sd[pre + "transl.weight"][idx]
        else:
            inp[f"{nid}.global_orient"] = sd[pre + "global_orient.weight"][idx]
            inp[f"{nid}.pose"] = sd[pre + "pose.weight"][idx]
            inp[f"{nid}.transl"] = sd[pre + "transl.weight"][idx]
            inp[f"{nid}.betas"] = sd[pre + "betas.weight"][torch.zeros_like(idx)]
    return b, inp


def hip_input(b, net, epoch=0, step=0):
    dev = torch.device("cuda:0")
    inp = {k: torch.from_numpy(v).to(dev) for k, v in b.items()}
    inp["current_epoch"], inp["global_step"] = epoch, step
    for node in net.nodes.values():
        inp.update(node.params(inp["idx"]))
    return inp


def rel_err(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-12))
