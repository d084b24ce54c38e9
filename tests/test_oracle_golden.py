"""CPU: the oracle restatement against fixtures recorded from the reference's own Python
(scripts/make_golden.py).  This is what pins oracle/hold_oracle.py."""
import os

import numpy as np
import torch

from parity_common import ho, oracle_input, setup, syn, z_in_reference_order  # noqa: F401

CFG = dict(W=8, H=8, frames_eval=[1, 3], frames_train=[0, 2])


def _load(gold_dir, name):
    return dict(np.load(os.path.join(gold_dir, name)))


def test_mano_server_matches_reference(gold_dir):
    g = _load(gold_dir, "mano.npz")
    sc, sd_np, sd, osc = setup()
    n = sc["n_frames"]
    pre = "nodes.right.params."
    full_pose = torch.cat([sd[pre + "global_orient.weight"], sd[pre + "pose.weight"]], 1)
    out = ho.mano_server(osc.mano["right"], osc.tfs_c_inv["right"], torch.full((n,), sc["scene_scale"]),
                         sd[pre + "transl.weight"], full_pose, sd[pre + "betas.weight"].expand(n, -1))
    for k in ["verts", "jnts", "tfs", "v_posed"]:
        assert np.abs(out[k].numpy() - g[k]).max() < 1e-6, k
    assert np.abs(osc.verts_c["right"].numpy() - g["verts_c"]).max() < 1e-6
    assert np.abs(osc.tfs_c_inv["right"].numpy() - g["tfs_c_inv"]).max() < 1e-5


def test_eval_forward_matches_reference_given_z(gold_dir):
    g = _load(gold_dir, "eval.npz")
    sc, sd_np, sd, osc = setup()
    b, inp = oracle_input(sc, sd, CFG["frames_eval"], CFG["W"], CFG["H"])
    zo = {n: torch.from_numpy(g[f"{n}.z_vals"]) for n in sc["entities"]}
    ex = {}
    out = ho.holdnet_forward(osc, sd, inp, False, z_override=zo, extras=ex)
    for k in ["rgb", "fg_rgb", "normal", "depth", "mask_prob", "semantics", "right.fg_rgb", "object.fg_rgb",
              "right.normal", "object.depth", "bg_rgb_only", "fg_weights", "bg_weights"]:
        assert np.abs(out[k].detach().numpy() - g["out." + k]).max() < 2e-5, k
    for n in sc["entities"]:
        assert np.abs(ex[n]["x_c"].detach().numpy() - g[f"{n}.x_c"].reshape(-1, 3)).max() < 1e-5
        assert np.abs(ex[n]["sdf"].detach().numpy() - g[f"{n}.sdf"].reshape(-1, 1)).max() < 2e-5
        assert np.abs(ex[n]["feat"].detach().numpy().sum(-1) - g[f"{n}.feat_sum"].reshape(-1)).max() < 2e-3


def test_eval_sampler_close_to_reference(gold_dir):
    g = _load(gold_dir, "eval.npz")
    sc, sd_np, sd, osc = setup()
    b, inp = oracle_input(sc, sd, CFG["frames_eval"], CFG["W"], CFG["H"])
    out = ho.holdnet_forward(osc, sd, inp, False)
    for n in sc["entities"]:
        dz = np.abs(out[f"{n}.z_vals"].numpy() - g[f"{n}.z_vals"])
        # inverse-CDF sampling is discontinuous: where u falls on a CDF step, fp32 reorderings put the sample into the
        # neighbouring bin (a jump of up to one coarse interval, ~0.06 here); that may hit a handful of samples
        assert (dz > 1e-4).mean() < 0.02 and dz.max() < 0.1, (n, dz.max(), (dz > 1e-4).mean())
    mse = ((out["rgb"].detach().numpy() - g["out.rgb"]) ** 2).mean()
    assert 10 * np.log10(1.0 / mse) > 60


def test_train_forward_backward_matches_reference(gold_dir):
    g = _load(gold_dir, "train.npz")
    sc, sd_np, sd, osc = setup()
    sdg = {k: (v.clone().requires_grad_(True) if v.dtype.is_floating_point else v) for k, v in sd.items()}
    b, inp = oracle_input(sc, sdg, CFG["frames_train"], CFG["W"], CFG["H"])
    nodes = list(sc["entities"].keys())
    rng = {"bg_t": torch.from_numpy(g[f"rand.{2 * len(nodes)}"])}
    for i, n in enumerate(nodes):
        rng[n] = {"t_uniform": torch.from_numpy(g[f"rand.{2 * i}"]), "u_final": torch.from_numpy(g[f"rand.{2 * i + 1}"]),
                  "perm": torch.from_numpy(g[f"perm.{i}"])}
    out = ho.holdnet_forward(osc, sdg, inp, True, rng=rng, current_epoch=25, barf_alpha_iter=4000)
    gt = torch.from_numpy(b["gt.rgb"]).view(-1, 3)
    loss = (out["rgb"] - gt).abs().mean() + 0.1 * (out["semantics"] ** 2).mean() + 0.05 * out["normal"].sum(-1).mean()
    assert abs(float(loss) - float(g["loss"])) < 2e-5
    assert np.abs(out["rgb"].detach().numpy() - g["out.rgb"]).max() < 1e-3
    loss.backward()
    # EVERY gradient tensor the reference recorded (dense layers, density betas, frame latents, pose / shape / translation
    # tables): relative to the reference tensor's norm.  Measured worst 1.3e-4 (fp32 summation order of
    # the two autograd graphs); the GPU suite holds the HIP path to 2e-4 of the oracle's fp64 autograd on the same tensors.
    names = [k[5:] for k in g if k.startswith("grad.")]
    assert len(names) >= 100, len(names)
    worst = ("", 0.0)
    for name in names:
        ref = g["grad." + name]
        assert sdg[name].grad is not None, name
        og = sdg[name].grad.numpy()
        if og.size > 4096:  # recorded as a strided sample of 1 024 elements (scripts/make_golden.py); gradnorm.* covers the whole
            og = og.reshape(-1)[:: max(1, og.size // 1024)][:1024]
        nr = np.linalg.norm(ref)
        if nr == 0.0:
            assert np.abs(og).max() == 0.0, name
            continue
        rel = np.linalg.norm(og - ref) / nr
        worst = max(worst, (name, rel), key=lambda t: t[1])
        assert rel < 3e-4, (name, rel)
    for name in [k[9:] for k in g if k.startswith("gradnorm.")]:
        gn = float(sdg[name].grad.norm())
        assert abs(gn - float(g["gradnorm." + name])) / float(g["gradnorm." + name]) < 3e-4, name
    print("worst gradient tensor vs the reference:", worst)


def _params_input(sc, sd, frames, W):
    b, inp = oracle_input(sc, sd, frames, W, W)
    return b, inp


def test_two_hand_eval_matches_reference(gold_dir):
    """configs[3]-like scene (right + left + object) against the reference's own run: pins the oracle's 3-node merge /
    trim (hold_utils.py:76-121) and the left-hand MANO server (mano/server.py:116-133)."""
    g = _load(gold_dir, "twohand_eval.npz")
    sc, sd_np, sd, osc = setup(n_frames=2, two_hands=True)
    b, inp = _params_input(sc, sd, [0, 1], 6)
    zo = {n: torch.from_numpy(g[f"{n}.z_vals"]) for n in sc["entities"]}
    ex = {}
    out = ho.holdnet_forward(osc, sd, inp, False, z_override=zo, extras=ex)
    assert out["fg_weights"].shape[1] == 3 * 98 - 2 * 3 + 1
    for k in [k[4:] for k in g if k.startswith("out.")]:
        if k in out and torch.is_tensor(out[k]) and out[k].dtype.is_floating_point:
            assert np.abs(out[k].detach().numpy() - g["out." + k]).max() < 2e-5, k
    for n in sc["entities"]:
        assert np.abs(ex[n]["x_c"].detach().numpy() - g[f"{n}.x_c"].reshape(-1, 3)).max() < 1e-5
        assert np.abs(ex[n]["sdf"].detach().numpy() - g[f"{n}.sdf"].reshape(-1, 1)).max() < 2e-5
    # ... and its own sampler lands on the reference's samples
    o2 = ho.holdnet_forward(osc, sd, inp, False)
    for n in sc["entities"]:
        dz = np.abs(o2[f"{n}.z_vals"].numpy() - g[f"{n}.z_vals"])
        assert (dz > 1e-4).mean() < 0.02 and dz.max() < 0.1, (n, dz.max())


def test_two_hand_train_matches_reference(gold_dir):
    """two-hand training step with the reference's recorded draws: loss and EVERY recorded gradient tensor (167)."""
    g = _load(gold_dir, "twohand_train.npz")
    sc, sd_np, sd, osc = setup(n_frames=2, two_hands=True)
    sdg = {k: (v.clone().requires_grad_(True) if v.dtype.is_floating_point else v) for k, v in sd.items()}
    b, inp = oracle_input(sc, sdg, [0, 1], 6, 6)
    nodes = ["right", "left", "object"]  # the reference's node order (hold_net.py: right, left, object)
    assert list(sc["entities"].keys()) == nodes
    rng = {"bg_t": torch.from_numpy(g[f"rand.{2 * len(nodes)}"])}
    for i, n in enumerate(nodes):
        rng[n] = {"t_uniform": torch.from_numpy(g[f"rand.{2 * i}"]), "u_final": torch.from_numpy(g[f"rand.{2 * i + 1}"]),
                  "perm": torch.from_numpy(g[f"perm.{i}"])}
    out = ho.holdnet_forward(osc, sdg, inp, True, rng=rng, current_epoch=25, barf_alpha_iter=4000)
    gt = torch.from_numpy(b["gt.rgb"]).view(-1, 3)
    loss = (out["rgb"] - gt).abs().mean() + 0.1 * (out["semantics"] ** 2).mean() + 0.05 * out["normal"].sum(-1).mean()
    assert abs(float(loss) - float(g["loss"])) < 2e-5
    loss.backward()
    names = [k[5:] for k in g if k.startswith("grad.")]
    assert len(names) >= 150, len(names)
    for name in names:
        ref = g["grad." + name]
        og = sdg[name].grad.numpy()
        if og.size > 4096:
            og = og.reshape(-1)[:: max(1, og.size // 1024)][:1024]
        nr = np.linalg.norm(ref)
        if nr == 0.0:
            assert np.abs(og).max() == 0.0, name
            continue
        assert np.linalg.norm(og - ref) / nr < 3e-4, (name, np.linalg.norm(og - ref) / nr)


def test_c1_c5_configs_match_reference(gold_dir):
    """BASELINE.json configs[0] (N_samples = 32, 64 x 64 rays) and configs[4] (N_samples = 128) of the reference itself:
    the oracle's own sampler against the reference's z_vals and, given those z, every recorded per-ray output."""
    from hold_amd import synthetic as syn
    mano = {"right": syn.make_mano_model(True), "left": syn.make_mano_model(False)}
    for name, ns, W, frames in [("c1_eval", 32, 64, [0]), ("c5_eval", 128, 16, [0, 1])]:
        g = _load(gold_dir, name + ".npz")
        sc, sd_np, sd, _ = setup()
        osc = ho.OracleScene(sc, mano, N_samples=ns)
        b, inp = _params_input(sc, sd, frames, W)
        zo = {n: torch.from_numpy(g[f"{n}.z_vals"]) for n in sc["entities"]}
        assert zo["right"].shape[1] == ns + 2 + 32
        if name == "c5_eval":  # C1 is 4 096 rays: its end-to-end oracle run (a minute on 8 cores) stays in make_golden_configs.py
            o2 = ho.holdnet_forward(osc, sd, inp, False)
            for n in sc["entities"]:
                dz = np.abs(o2[f"{n}.z_vals"].numpy() - g[f"{n}.z_vals"])
                assert (dz > 1e-4).mean() < 0.02 and dz.max() < 0.1, (name, n, dz.max())
            zo_run = zo
        else:  # a 256-ray subset of the 4 096 (rays are independent given the weights)
            sel = torch.arange(0, 4096, 16)
            inp = {k: (v[:, sel] if (torch.is_tensor(v) and v.dim() >= 2 and v.shape[1] == 4096) else v) for k, v in inp.items()}
            zo_run = {n: z[sel] for n, z in zo.items()}
        out = ho.holdnet_forward(osc, sd, inp, False, z_override=zo_run)
        n_cmp = 0
        for k in [k[4:] for k in g if k.startswith("out.")]:
            if k in out and torch.is_tensor(out[k]) and out[k].dtype.is_floating_point and k not in ("ray_dirs", "cam_loc"):
                ref = g["out." + k]
                if name == "c1_eval":
                    ref = ref[sel.numpy()] if ref.shape[0] == 4096 else ref
                if ref.shape != tuple(out[k].shape):
                    continue
                assert np.abs(out[k].detach().numpy() - ref).max() < 2e-5, (name, k)
                n_cmp += 1
        assert n_cmp >= 15, (name, n_cmp)


def test_reference_tie_order_made_explicit_in_z_reproduces_the_reference_composite(gold_dir):
    """parity_common.z_in_reference_order: the fixtures' z with every group of equal values spread by a few ulps in the order
    the reference's unstable torch.sort put them.  A STABLE merge of that z (the HIP compositor's, and the oracle's
    stable_merge mode) must then reproduce the reference's own composite outputs -- every key, every ray -- which is what lets
    the GPU tests hold the HIP composite to the reference fixtures directly.  Also: the tie order matters on EVERY ray (so
    excluding tie rays, as VERDICT r5 suggested, would leave nothing to test)."""
    g = _load(gold_dir, "eval.npz")
    sc, sd_np, sd, osc = setup()
    b, inp = oracle_input(sc, sd, CFG["frames_eval"], CFG["W"], CFG["H"])
    nodes = list(sc["entities"])
    keys = ["rgb", "fg_rgb", "normal", "depth", "mask_prob", "semantics", "fg_weights", "bg_weights", "right.depth", "object.normal"]
    plain = ho.holdnet_forward(osc, sd, inp, False, z_override={n: torch.from_numpy(g[f"{n}.z_vals"]) for n in nodes}, stable_merge=True)
    moved = np.stack([np.abs(plain[k].detach().numpy() - g["out." + k]).reshape(len(g["out.rgb"]), -1).max(1) for k in keys[:6]]).max(0)
    assert (moved > 1e-5).all() and moved.max() > 5e-3  # earlier-node-first is NOT the reference's order, on every ray
    zo = z_in_reference_order(g, nodes)
    for n in nodes:
        assert float((zo[n] - torch.from_numpy(g[f"{n}.z_vals"])).abs().max()) < 1e-5
    out = ho.holdnet_forward(osc, sd, inp, False, z_override=zo, stable_merge=True)
    for k in keys:
        assert np.abs(out[k].detach().numpy() - g["out." + k]).max() < 2e-5, k
    g2 = _load(gold_dir, "twohand_eval.npz")
    sc2, _, sd2, osc2 = setup(n_frames=2, two_hands=True)
    b2, inp2 = _params_input(sc2, sd2, [0, 1], 6)
    out2 = ho.holdnet_forward(osc2, sd2, inp2, False, z_override=z_in_reference_order(g2, list(sc2["entities"])), stable_merge=True)
    for k in keys[:8] + ["left.depth"]:
        assert np.abs(out2[k].detach().numpy() - g2["out." + k]).max() < 2e-5, k


def test_beta_search_conditioning_on_the_reference_trace(gold_dir):
    """oracle.beta_search_conditioning (the yardstick of the GPU beta-search test): on the reference-pinned sampler trace the
    fp64 search and the recorded fp32 betas agree to 1e-4 on every ray the fp64 run calls well conditioned, almost every ray is,
    and a wider noise band only ever ADDS borderline rays."""
    g = _load(gold_dir, "sampler.npz")
    beta0 = 0.1 + 1e-4
    for r in range(int(g["n_rounds"])):
        z, sdf = torch.from_numpy(g[f"r{r}.z_vals"]), torch.from_numpy(g[f"r{r}.sdf"])
        b_in = (torch.sqrt((1.0 / (4.0 * np.log(1.1))) * ((z[:, 1:] - z[:, :-1]) ** 2).sum(-1)) if r == 0
                else torch.from_numpy(g[f"r{r - 1}.beta"]))
        ref = torch.from_numpy(g[f"r{r}.beta"]).double()
        b64, bl, margin = ho.beta_search_conditioning(z, sdf, b_in, beta0, 0.1, 10, 1e-4)
        assert int(bl.sum()) <= 2 and bool(((ref - b64).abs()[~bl] <= 1e-4 * b64[~bl]).all())
        _, bl3, _ = ho.beta_search_conditioning(z, sdf, b_in, beta0, 0.1, 10, 1e-3)
        assert bool((bl3 | ~bl).all()) and bool((margin >= 0).all())
        b32 = ho.sampler_round(z, sdf, b_in.clone(), beta0, 0.1, 10)[0]
        assert torch.equal(b32, ref.float())  # the fp32 oracle IS the recorded trace


def test_oracle_replays_the_reference_modules_three_training_steps(gold_dir):
    """tests/golden/hold_steps.npz = the reference's Lightning module at work (scripts/make_golden_hold_steps.py:
    HOLD.training_step x 3 with its Loss, configure_optimizers' Adam, clip_grad_norm_(0.5)).  The oracle, fed the recorded
    draws, its OWN sampler in the loop, its Loss restatement, autograd, the same clip and torch.optim.Adam with the reference's
    groups, must walk the same trajectory: per-step loss and gradient norm, z_vals up to the discontinuous inverse-CDF stage,
    every parameter after the third update.  (The GPU suite replays the same recording through the HIP path:
    tests/test_dropin_gpu.py.)"""
    from oracle import targets_oracle as to
    g = _load(gold_dir, "hold_steps.npz")
    sc, sd_np, sd, osc = setup(n_frames=4, barf_iter=int(g["cfg.barf_iter"]))
    nodes = list(sc["entities"])
    sdg = {k: (v.clone().requires_grad_(True) if v.dtype.is_floating_point else v.clone()) for k, v in sd.items()}
    names = [k[2:] for k in g if k.startswith("p.")]
    # (the MANO layer's own pose / shape parameters, body_models.py, are trainable by flag but never reached by a gradient)
    assert all(float(g["dnorm." + n]) == 0.0 for n in names if n not in sdg)
    names = [n for n in names if n in sdg]
    node_names = [n for n in names if ".params." in n]
    lr = float(g["cfg.lr"])
    opt = torch.optim.Adam([{"params": [sdg[n] for n in node_names], "lr": 0.1 * lr},
                            {"params": [sdg[n] for n in names if n not in node_names], "lr": lr}], lr=lr, eps=1e-8)
    W, epoch = int(g["cfg.W"]), int(g["cfg.epoch"])
    uv = syn.make_uv(W, W)
    for k in range(int(g["cfg.steps"])):
        pre = f"s{k}."
        step = int(g["cfg.first_step"]) + k
        b = syn.make_batch(sc, g["cfg.frames"][k].tolist(), uv, W, W, seed=1 + k)
        inp = {kk: torch.from_numpy(v) for kk, v in b.items()}
        idx = inp["idx"]
        for nid in nodes:
            p_ = f"nodes.{nid}.params."
            inp[f"{nid}.global_orient"], inp[f"{nid}.transl"] = sdg[p_ + "global_orient.weight"][idx], sdg[p_ + "transl.weight"][idx]
            if nid != "object":
                inp[f"{nid}.pose"], inp[f"{nid}.betas"] = sdg[p_ + "pose.weight"][idx], sdg[p_ + "betas.weight"][torch.zeros_like(idx)]
        rng = {"bg_t": torch.from_numpy(g[pre + f"rand.{2 * len(nodes)}"])}
        for i, nid in enumerate(nodes):
            rng[nid] = {"t_uniform": torch.from_numpy(g[pre + f"rand.{2 * i}"]), "u_final": torch.from_numpy(g[pre + f"rand.{2 * i + 1}"]),
                        "perm": torch.from_numpy(g[pre + f"perm.{i}"])}
        # the oracle's own sampler with the recorded draws ...
        # (detached copies, not no_grad: the oracle's normal path takes d sdf / d x through autograd)
        o_s = ho.holdnet_forward(osc, {kk: v.detach() for kk, v in sdg.items()}, {kk: (v.detach() if torch.is_tensor(v) else v) for kk, v in inp.items()},
                                 True, rng=rng, current_epoch=epoch, barf_alpha_iter=int(g["cfg.barf_iter"]) + 1 + k) if k == 0 else None
        if o_s is not None:
            for nid in nodes:
                dz = np.abs(o_s[f"{nid}.z_vals"].numpy() - g[pre + f"{nid}.z_vals"])
                assert (dz > 1e-4).mean() < 0.02 and dz.max() < 0.1, (nid, dz.max())
        # ... and the step itself on the reference's z (a moved sample would move the loss by more than the bar below)
        zo = {nid: torch.from_numpy(g[pre + f"{nid}.z_vals"]) for nid in nodes}
        out = ho.holdnet_forward(osc, sdg, inp, True, rng=rng, z_override=zo, current_epoch=epoch,
                                 barf_alpha_iter=int(g["cfg.barf_iter"]) + 1 + k)
        out["step"], out["epoch"] = step, epoch
        ld = to.loss_forward({"gt.rgb": inp["gt.rgb"], "gt.mask": inp["gt.mask"]}, out)
        assert abs(float(ld["loss"]) - float(g[pre + "loss"])) < 1e-5 * float(g[pre + "loss"]), (k, float(ld["loss"]), float(g[pre + "loss"]))
        for key in ("rgb", "semantics", "depth", "mask_prob"):
            assert np.abs(out[key].detach().numpy().reshape(g[pre + "out." + key].shape) - g[pre + "out." + key]).max() < 5e-5, (k, key)
        opt.zero_grad()
        ld["loss"].backward()
        gn = torch.nn.utils.clip_grad_norm_([sdg[n] for n in names], float(g["cfg.clip"]))
        assert abs(float(gn) - float(g[pre + "grad_norm"])) < 3e-4 * float(g[pre + "grad_norm"]), (k, float(gn), float(g[pre + "grad_norm"]))
        opt.step()
    worst = ("", 0.0)
    for n in names:
        t = sdg[n].detach().reshape(-1)
        t = t if t.numel() <= 4096 else t[:: max(1, t.numel() // 1024)][:1024]
        share = np.sqrt(t.numel() / sdg[n].numel())
        r = np.linalg.norm(t.double().numpy() - g["p." + n]) / (float(g["pnorm." + n]) * share + 1e-30)
        worst = max(worst, (n, r), key=lambda x: x[1])
    assert worst[1] < 1e-4, worst


def test_fitting_losses_match_reference(gold_dir):
    """oracle/fitting_oracle.py:loss_fn_h / loss_fn_ih against the reference's own code/src/fitting/loss.py outputs"""
    from fitting_loss_cases import run_single_hand, run_two_hand
    from oracle import fitting_oracle as fo
    g = _load(gold_dir, "fitting_losses.npz")
    assert run_two_hand(g, fo.loss_fn_ih, "cpu") < 1e-6
    assert run_single_hand(g, fo.loss_fn_h, "cpu") < 1e-6


def test_geometry_oracle_matches_closed_form_sdf():
    """oracle/geometry_oracle.py (point->mesh distance + winding-number sign, the kaolin replacement of SURVEY 8(f-2))
    against the closed-form signed distance of a box and the half-space inside test of a tetrahedron."""
    from oracle import geometry_oracle as go
    g = torch.Generator().manual_seed(0)
    h = (0.3, 0.2, 0.5)
    v, f = go.box_mesh(h)
    p = (torch.rand(4000, 3, generator=g, dtype=torch.float64) * 2 - 1) * 0.8
    ref = go.box_sdf(p, h)
    assert float((go.mesh_sdf(p, v, f) - ref).abs().max()) < 1e-12
    assert float((go.mesh_sdf(p.float(), v.float(), f).double() - ref).abs().max()) < 1e-6
    tv = torch.tensor([[0., 0, 0], [1, 0, 0], [0, 1, 0], [0, 0, 1]], dtype=torch.float64)
    tf = torch.tensor([[0, 2, 1], [0, 1, 3], [0, 3, 2], [1, 2, 3]])
    pt = torch.rand(2000, 3, generator=g, dtype=torch.float64) * 1.4 - 0.2
    inside = (pt.min(1).values > 0) & (pt.sum(1) < 1)
    assert torch.equal(go.mesh_sdf(pt, tv, tf) < 0, inside)
    # per-ray masks of check_off_in_surface_points_cano_mesh (volsdf_utils.py:189-217): 10 rays x 8 samples
    x = p[:80].reshape(1, 80, 3)
    off, ins = go.check_off_in_surface_points_cano_mesh(v[None], f, x, 10, threshold=0.05)
    m = ref[:80].reshape(10, 8).min(1).values
    assert torch.equal(off, m > 0.05) and torch.equal(ins, m <= 0.0)
