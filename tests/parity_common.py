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


def z_in_reference_order(g, nodes):
    """The fixture's per-node z_vals with every group of EQUAL values spread by a few ulps in the order the reference's own call
    put them.  merge_factors (code/src/hold/hold_utils.py:76-121) sorts cat(z of all nodes) WITHOUT `stable`; equal z exist on
    every ray (near = 0 and the sphere exit in every node, and in eval mode the extras repeat the uniform samples all nodes
    share), their order is an artefact of torch's unstable sort, and which node's sample gets the interval behind a tie moves
    depth / normal / semantics of a ray by up to 1e-2 -- EVERY ray of every eval fixture, measured.  The fixtures were
    recorded with this torch build, so the same call reproduces the reference's permutation (tests/test_oracle_golden.py:
    the oracle with that call == the fixtures to 4e-7); spreading a tie group by rank x ulp(z) (1e-10 at z = 0) makes the
    order explicit in the DATA: a stable merge of the result IS the reference's merge, and the HIP compositor can be held to
    the reference's composite outputs directly.  z moves by <= 5e-6."""
    Z = torch.cat([torch.from_numpy(np.asarray(g[f"{n}.z_vals"])) for n in nodes], 1)
    S = Z.shape[1] // len(nodes)
    zs, idx = torch.sort(Z, dim=1)  # the reference's call
    rank = torch.zeros_like(idx)
    for j in range(1, zs.shape[1]):
        rank[:, j] = torch.where(zs[:, j] == zs[:, j - 1], rank[:, j - 1] + 1, torch.zeros_like(rank[:, j]))
    ulp = torch.maximum(zs.abs() * 2.0 ** -23, torch.full_like(zs, 1e-10)).double()
    zn = (zs.double() + rank.double() * ulp).float()
    assert bool((zn[:, 1:] > zn[:, :-1]).all()), "spread z not strictly increasing"
    assert float((zn - zs).abs().max()) < 1e-5
    out = torch.empty_like(Z)
    out.scatter_(1, idx, zn)
    # within a node equal z are the same point with the same factors: each node's list is simply sorted again
    return {n: torch.sort(out[:, i * S:(i + 1) * S], dim=1).values.contiguous() for i, n in enumerate(nodes)}
