"""More fixtures from the REFERENCE's own Python (this container only), for the configurations VERDICT r3 found pinned to
the oracle alone: the two-hand scene (configs[3]: right + left + object -- the 3-node merge with its [(n-1) : -n] trim,
code/src/hold/hold_utils.py:76-121, and the left-hand server, code/src/model/mano/server.py:116-133), eval and train, and
the sampler configurations of configs[0] (N_samples = 32) and configs[4] (N_samples = 128).

    python scripts/make_golden_configs.py     # writes tests/golden/{twohand_eval,twohand_train,c1_eval,c5_eval}.npz

Inputs are regenerated from seeds by hold_amd.synthetic (tests/parity_common.py), so the fixtures only hold OUTPUTS, the
reference's own z_vals and the random draws of the training run.  Prints the oracle's deviation from each fixture.
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "scripts"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from hold_amd import synthetic as syn  # noqa: E402
from oracle import hold_oracle as ho  # noqa: E402
from oracle import ref_shim  # noqa: E402
from make_golden import RandRecorder, capture_nodes, np_, to_input  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
# (name, two_hands, n_frames, frames, W, N_samples)
EVAL_CASES = [("twohand_eval", True, 2, [0, 1], 6, None), ("c1_eval", False, 4, [0], 64, 32), ("c5_eval", False, 4, [0, 1], 16, 128)]
TRAIN_CASE = ("twohand_train", True, 2, [0, 1], 6)


def build(two_hands, n_frames, n_samples):
    sc = syn.make_scene(n_frames=n_frames, two_hands=two_hands)
    net, opt, args, wd = ref_shim.build_holdnet(sc, perturb=0, sampler=None if n_samples is None else {"N_samples": n_samples})
    sd_np = syn.make_state_dict(sc, barf_iter=3999)
    missing, unexpected = net.load_state_dict({k: torch.as_tensor(v) for k, v in sd_np.items()}, strict=False)
    assert not unexpected, unexpected
    for node in net.nodes.values():
        node.params.defrost()
        node.implicit_network.embedder_obj.step()
    return sc, net, sd_np


def oracle_inputs(sc, sd, b):
    inp = {k: torch.from_numpy(v) for k, v in b.items()}
    idx = inp["idx"]
    for nid in sc["entities"]:
        pre = f"nodes.{nid}.params."
        if nid == "object":
            inp["object.global_orient"], inp["object.transl"] = sd[pre + "global_orient.weight"][idx], sd[pre + "transl.weight"][idx]
        else:
            inp[f"{nid}.global_orient"], inp[f"{nid}.pose"] = sd[pre + "global_orient.weight"][idx], sd[pre + "pose.weight"][idx]
            inp[f"{nid}.transl"] = sd[pre + "transl.weight"][idx]
            inp[f"{nid}.betas"] = sd[pre + "betas.weight"][torch.zeros_like(idx)]
    return inp


def main():
    os.makedirs(GOLD, exist_ok=True)
    torch.set_num_threads(os.cpu_count())
    mano = {"right": syn.make_mano_model(True), "left": syn.make_mano_model(False)}
    for name, two, nf, frames, W, ns in EVAL_CASES:
        sc, net, sd_np = build(two, nf, ns)
        net.eval()
        for node in net.nodes.values():
            node.implicit_network.embedder_obj.eval()
        uv = syn.make_uv(W, W)
        b = syn.make_batch(sc, frames, uv, W, W)
        with torch.no_grad() if False else torch.enable_grad():
            out = net(to_input(b, net))
        cap = capture_nodes(net, to_input(b, net))
        # the two-hand fixture keeps the per-sample intermediates; the 4 096- / 512-ray sampler configurations keep the
        # reference's z_vals and the per-ray outputs only (the per-sample arrays and weight matrices would be 30 MB)
        full = ns is None
        gold = {f"out.{k}": np_(v) for k, v in out.items()
                if torch.is_tensor(v) and (full or "fg_weights" not in k) and (full or k == "bg_weights" or "bg_weights" not in k)}
        for nid, rec in cap.items():
            for k in (("z_vals", "x_c", "sdf", "color", "normal", "density", "verts", "tfs") if full else ("z_vals", "verts")):
                if k in rec:
                    gold[f"{nid}.{k}"] = np_(rec[k])
        np.savez_compressed(os.path.join(GOLD, name + ".npz"), **gold)
        # the oracle on the same inputs: (1) end to end (its own sampler), (2) with the reference's z fed in
        sd = {k: torch.as_tensor(v) for k, v in sd_np.items()}
        osc = ho.OracleScene(sc, mano) if ns is None else ho.OracleScene(sc, mano, N_samples=ns)
        oinp = oracle_inputs(sc, sd, b)
        zo = {n: cap[n]["z_vals"].detach() for n in cap}
        oo = ho.holdnet_forward(osc, sd, oinp, False, z_override=zo)
        o2 = ho.holdnet_forward(osc, sd, oinp, False)
        print(f"== {name}: {len(frames) * W * W} rays, z per node {cap[next(iter(cap))]['z_vals'].shape[1]}; oracle vs reference (max abs)")
        for k in sorted(gold):
            if k.startswith("out.") and k[4:] in oo and torch.is_tensor(oo[k[4:]]) and oo[k[4:]].dtype.is_floating_point:
                print(f"   {k[4:]:22s} given z {float((oo[k[4:]] - out[k[4:]]).abs().max()):.3e}")
        for n in cap:
            dz = (o2[n + ".z_vals"] - cap[n]["z_vals"]).abs()
            print(f"   {n}.z_vals own sampler: max {float(dz.max()):.3e}, moved > 1e-4: {float((dz > 1e-4).float().mean()):.4f}")

    # ---------------- two-hand training step: fwd + bwd with recorded draws ----------------
    name, two, nf, frames, W = TRAIN_CASE
    sc, net, sd_np = build(two, nf, None)
    net.train()
    for node in net.nodes.values():
        node.implicit_network.embedder_obj.no_barf = False
    uv = syn.make_uv(W, W)
    b = syn.make_batch(sc, frames, uv, W, W)
    torch.manual_seed(13)
    with RandRecorder() as rr:
        out = net(to_input(b, net, epoch=25, step=10))
    gt = torch.from_numpy(b["gt.rgb"]).view(-1, 3)
    loss = (out["rgb"] - gt).abs().mean() + 0.1 * (out["semantics"] ** 2).mean() + 0.05 * out["normal"].sum(-1).mean()
    net.zero_grad()
    loss.backward()
    grads = {n: p.grad.detach().clone() for n, p in net.named_parameters() if p.grad is not None}
    gold = {f"out.{k}": np_(v) for k, v in out.items() if torch.is_tensor(v)}
    gold["loss"] = np_(loss)
    for i, r in enumerate(rr.rand):
        gold[f"rand.{i}"] = np_(r)
    for i, r in enumerate(rr.perm):
        gold[f"perm.{i}"] = np_(r)
    for n, g in grads.items():
        gold[f"gradnorm.{n}"] = np_(g.norm())
        gold[f"grad.{n}"] = np_(g) if g.numel() <= 4096 else np_(g.reshape(-1)[:: max(1, g.numel() // 1024)][:1024])
    np.savez_compressed(os.path.join(GOLD, name + ".npz"), **gold)
    nodes = list(net.nodes.keys())
    assert len(rr.rand) == 2 * len(nodes) + 1 and len(rr.perm) == len(nodes), (len(rr.rand), len(rr.perm), nodes)
    rng = {"bg_t": rr.rand[-1]}
    for i, nid in enumerate(nodes):
        rng[nid] = {"t_uniform": rr.rand[2 * i], "u_final": rr.rand[2 * i + 1], "perm": rr.perm[i]}
    sd = {k: torch.as_tensor(v) for k, v in sd_np.items()}
    sdg = {k: (v.clone().requires_grad_(True) if v.dtype.is_floating_point else v) for k, v in sd.items()}
    osc = ho.OracleScene(sc, mano)
    oo = ho.holdnet_forward(osc, sdg, oracle_inputs(sc, sdg, b), True, rng=rng, current_epoch=25, barf_alpha_iter=4000)
    oloss = (oo["rgb"] - gt).abs().mean() + 0.1 * (oo["semantics"] ** 2).mean() + 0.05 * oo["normal"].sum(-1).mean()
    oloss.backward()
    worst = ("", 0.0)
    for n, g in grads.items():
        og = sdg[n].grad
        if og is None:
            print("  oracle has no grad for", n)
            continue
        rel = float((og - g).norm() / (g.norm() + 1e-12))
        worst = max(worst, (n, rel), key=lambda t: t[1])
    print(f"== {name}: node order {nodes}; loss {float(loss):.6f} vs oracle {float(oloss):.6f}; rgb max abs "
          f"{float((oo['rgb'] - out['rgb']).abs().max()):.3e}; {len(grads)} gradient tensors, worst rel {worst}")


if __name__ == "__main__":
    main()
