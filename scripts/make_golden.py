"""Generate tests/golden/*.npz by running the REFERENCE's own Python on CPU (this container only;
/root/reference is absent on the GPU box).  Also prints oracle-vs-reference deviations so the
restatement in oracle/hold_oracle.py stays pinned.

    python scripts/make_golden.py            # writes tests/golden/{eval,train,mano,sampler}.npz

Inputs are regenerated from seeds by hold_amd.synthetic, so fixtures only hold OUTPUTS.
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from hold_amd import synthetic as syn  # noqa: E402
from oracle import hold_oracle as ho  # noqa: E402
from oracle import ref_shim  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
CFG = dict(n_frames=4, W=8, H=8, frames_eval=[1, 3], frames_train=[0, 2], barf_iter=3999)


def to_input(b, net, epoch=0, step=0):
    from common.xdict import xdict

    inp = xdict({k: torch.from_numpy(v) for k, v in b.items()})
    inp["current_epoch"] = epoch
    inp["global_step"] = step
    for node in net.nodes.values():
        inp.update(node.params(inp["idx"]))
    return inp


class RandRecorder:
    """records torch.rand / torch.randperm draws made by the reference, in call order."""

    def __enter__(self):
        self.rand, self.perm = [], []
        self._r, self._p = torch.rand, torch.randperm

        def rand(*a, **k):
            o = self._r(*a, **k)
            self.rand.append(o.clone())
            return o

        def randperm(*a, **k):
            o = self._p(*a, **k)
            self.perm.append(o.clone())
            return o

        torch.rand, torch.randperm = rand, randperm
        return self

    def __exit__(self, *a):
        torch.rand, torch.randperm = self._r, self._p


def build():
    sc = syn.make_scene(n_frames=CFG["n_frames"])
    net, opt, args, wd = ref_shim.build_holdnet(sc, perturb=0)
    sd_np = syn.make_state_dict(sc, barf_iter=CFG["barf_iter"])
    net.load_state_dict({k: torch.as_tensor(v) for k, v in sd_np.items()}, strict=False)
    for node in net.nodes.values():
        node.params.defrost()
        node.implicit_network.embedder_obj.step()  # refresh BARF weights from the loaded alpha_iter
    return sc, net, sd_np


def capture_nodes(net, inp):
    """run each node by itself to record intermediates (Node.forward returns sample_dict)."""
    import src.engine.volsdf_utils as vu

    cap = {}
    orig = vu.sdf_func_with_deformer
    for nid, node in net.nodes.items():
        rec = {}

        def patched(deformer, sdf_fn, training, x, deform_info, _rec=rec):
            o = orig(deformer, sdf_fn, training, x, deform_info)
            if torch.is_grad_enabled():
                _rec["sdf"], _rec["x_c"], _rec["feat"] = o
            return o

        vu.sdf_func_with_deformer = patched
        try:
            factors, sd_ = node(inp)
        finally:
            vu.sdf_func_with_deformer = orig
        rec.update(z_vals=sd_["z_vals"], color=factors["color"], normal=factors["normal"],
                   density=factors["density"], tfs=sd_["tfs"])
        if "output" in sd_:
            rec.update(verts=sd_["output"]["verts"], jnts=sd_["output"]["jnts"])
        cap[nid] = rec
    return cap


def np_(t):
    return t.detach().cpu().numpy()


def main():
    os.makedirs(GOLD, exist_ok=True)
    torch.set_num_threads(os.cpu_count())
    sc, net, sd_np = build()
    mano_models = {"right": syn.make_mano_model(True), "left": syn.make_mano_model(False)}
    osc = ho.OracleScene(sc, mano_models)
    sd = {k: torch.as_tensor(v) for k, v in sd_np.items()}
    uv = syn.make_uv(CFG["W"], CFG["H"])

    # ---------------- eval ----------------
    net.eval()
    for node in net.nodes.values():
        node.implicit_network.embedder_obj.eval()  # render.py:43-47
    b = syn.make_batch(sc, CFG["frames_eval"], uv, CFG["W"], CFG["H"])
    inp = to_input(b, net)
    out = net(inp)
    cap = capture_nodes(net, to_input(b, net))
    gold = {f"out.{k}": np_(v) for k, v in out.items() if torch.is_tensor(v)}
    for nid, rec in cap.items():
        for k, v in rec.items():
            if k == "feat":
                gold[f"{nid}.feat_head"] = np_(v)[..., :8]
                gold[f"{nid}.feat_sum"] = np_(v).sum(-1)
            else:
                gold[f"{nid}.{k}"] = np_(v)
    np.savez_compressed(os.path.join(GOLD, "eval.npz"), **gold)

    oinp = {k: torch.from_numpy(v) for k, v in b.items()}
    for node in net.nodes.values():
        oinp.update({k: v.detach() for k, v in node.params(oinp["idx"]).items()})
    ex = {}
    oo = ho.holdnet_forward(osc, sd, oinp, False, extras=ex)
    print("== eval: oracle vs reference (max abs) ==")
    for k in ["rgb", "fg_rgb", "normal", "depth", "mask_prob", "bg_rgb_only", "semantics", "right.fg_rgb",
              "object.fg_rgb", "fg_weights"]:
        print(f"  {k:14s} {float((oo[k] - out[k]).abs().max()):.3e}")
    for nid in cap:
        print(f"  {nid}.z_vals    {float((oo[nid + '.z_vals'] - cap[nid]['z_vals']).abs().max()):.3e}  iters={ex[nid]['iters']}")
        print(f"  {nid}.x_c       {float((ex[nid]['x_c'] - cap[nid]['x_c'].reshape(-1, 3)).abs().max()):.3e}")
        print(f"  {nid}.sdf       {float((ex[nid]['sdf'] - cap[nid]['sdf'].reshape(-1, 1)).abs().max()):.3e}")

    # ---------------- MANO server for all frames (a9) ----------------
    node = net.nodes["right"]
    idx = torch.arange(sc["n_frames"])
    p = node.params(idx)
    so = node.server(torch.full((sc["n_frames"],), sc["scene_scale"]), p["right.transl"], p["right.full_pose"],
                     p["right.betas"])
    np.savez_compressed(os.path.join(GOLD, "mano.npz"), verts=np_(so["verts"]), jnts=np_(so["jnts"]),
                        tfs=np_(so["tfs"]), v_posed=np_(so["v_posed"]), verts_c=np_(node.server.verts_c),
                        tfs_c_inv=np_(node.server.tfs_c_inv), deformer_verts=np_(node.deformer.verts))
    oso = ho.mano_server(osc.mano["right"], osc.tfs_c_inv["right"], torch.full((sc["n_frames"],), sc["scene_scale"]),
                         p["right.transl"], p["right.full_pose"], p["right.betas"])
    print("== mano: verts %.3e tfs %.3e jnts %.3e verts_c %.3e" % (
        float((oso["verts"] - so["verts"]).abs().max()), float((oso["tfs"] - so["tfs"]).abs().max()),
        float((oso["jnts"] - so["jnts"]).abs().max()), float((osc.verts_c["right"] - node.server.verts_c).abs().max())))

    # ---------------- sampler trace (a5) from the oracle run, pinned by the z_vals match above ----------
    tr = ex["right"]["trace"]
    np.savez_compressed(os.path.join(GOLD, "sampler.npz"),
                        **{f"r{i}.{k}": np_(v) for i, t in enumerate(tr) for k, v in t.items() if torch.is_tensor(v)},
                        z_final=np_(oo["right.z_vals"]), n_rounds=len(tr))

    # ---------------- train (fwd + bwd of L1 rgb + semantic CE-free proxy) ----------------
    net.train()
    for node in net.nodes.values():
        node.implicit_network.embedder_obj.no_barf = False
    b = syn.make_batch(sc, CFG["frames_train"], uv, CFG["W"], CFG["H"])
    torch.manual_seed(11)
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
        if g.numel() <= 4096:
            gold[f"grad.{n}"] = np_(g)
        else:
            gold[f"grad.{n}"] = np_(g.reshape(-1)[:: max(1, g.numel() // 1024)][:1024])
    np.savez_compressed(os.path.join(GOLD, "train.npz"), **gold)

    # oracle in training mode with the recorded draws
    Nr = len(CFG["frames_train"]) * uv.shape[0]
    nodes = list(net.nodes.keys())
    assert len(rr.rand) == 2 * len(nodes) + 1 and len(rr.perm) == len(nodes), (len(rr.rand), len(rr.perm))
    rng = {"bg_t": rr.rand[-1]}
    for i, nid in enumerate(nodes):
        rng[nid] = {"t_uniform": rr.rand[2 * i], "u_final": rr.rand[2 * i + 1], "perm": rr.perm[i]}
    sdg = {k: (v.clone().requires_grad_(True) if v.dtype.is_floating_point else v) for k, v in sd.items()}
    oinp = {k: torch.from_numpy(v) for k, v in b.items()}
    for nid in nodes:
        pre = f"nodes.{nid}.params."
        idxs = oinp["idx"]
        if nid == "object":
            oinp["object.global_orient"] = sdg[pre + "global_orient.weight"][idxs]
            oinp["object.transl"] = sdg[pre + "transl.weight"][idxs]
        else:
            oinp[f"{nid}.global_orient"] = sdg[pre + "global_orient.weight"][idxs]
            oinp[f"{nid}.pose"] = sdg[pre + "pose.weight"][idxs]
            oinp[f"{nid}.transl"] = sdg[pre + "transl.weight"][idxs]
            oinp[f"{nid}.betas"] = sdg[pre + "betas.weight"][torch.zeros_like(idxs)]
    oo = ho.holdnet_forward(osc, sdg, oinp, True, rng=rng, current_epoch=25, barf_alpha_iter=CFG["barf_iter"] + 1)
    oloss = (oo["rgb"] - gt).abs().mean() + 0.1 * (oo["semantics"] ** 2).mean() + 0.05 * oo["normal"].sum(-1).mean()
    oloss.backward()
    print("== train: oracle vs reference ==")
    print(f"  loss {float(loss):.6f} vs {float(oloss):.6f}; rgb max abs {float((oo['rgb'] - out['rgb']).abs().max()):.3e}")
    worst = 0.0
    for n, g in grads.items():
        og = sdg[n].grad
        if og is None:
            print("  oracle has no grad for", n)
            continue
        rel = float((og - g).norm() / (g.norm() + 1e-12))
        worst = max(worst, rel)
        if rel > 1e-3 or "params" in n or "beta" in n:
            print(f"  {n:60s} rel {rel:.3e} |g| {float(g.norm()):.3e}")
    print("  worst rel grad err", worst)


if __name__ == "__main__":
    main()
