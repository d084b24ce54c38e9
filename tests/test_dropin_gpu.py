"""The reference's own callers, replayed on the MI355X (VERDICT r5 missing #1).

The reference tree cannot travel to the GPU box in any form, so what travels is a RECORDING of its callers at work:
scripts/make_golden_hold_steps.py builds the reference's Lightning module ``src.hold.hold.HOLD(opt, args)`` -- the reference's
own HOLDNet inside -- on CPU, takes its ``configure_optimizers()`` Adam, runs ``HOLD.training_step`` three times with the
reference's ``Loss``, ``clip_grad_norm_(0.5)`` and ``optimizer.step()`` (code/src/hold/hold.py:79-137, code/train.py:28-73) and
then ``HOLD.inference_step`` on a 64 x 64 frame in 512-pixel chunks (hold.py:169-208), and stores losses, gradient norms,
z_vals, random draws, outputs, every parameter after the third step and the merged frame (tests/golden/hold_steps.npz).

Here the same three steps and the same frame go through hold_amd's side of that boundary -- ``hold_amd.train.training_step``
/ ``inference_step`` (the Lightning-free mirror of the two methods), ``hold_amd.loss.Loss``, ``FlatAdam`` (the mirror of the
optimiser + clip) around the HIP HOLDNet -- and are held to the recording: per-step loss 1e-5, outputs 1e-4, every parameter
within 1e-4 of its norm after three updates, per-node render keys 1e-4 and rgb PSNR > 50 dB for the frame.  The reference's
z_vals are fed in (the sampler's own parity, incl. its end-to-end run against these very z, is the last test below), with
the reference's order of equal z made explicit (parity_common.z_in_reference_order)."""
import os

import numpy as np
import pytest
import torch

from parity_common import setup, syn, z_in_reference_order

pytestmark = pytest.mark.gpu


def _sample(t):
    t = t.detach().reshape(-1)
    return t if t.numel() <= 4096 else t[:: max(1, t.numel() // 1024)][:1024]


def _net(g, train):
    import hold_amd
    sc, sd_np, sd, osc = setup(n_frames=4, barf_iter=int(g["cfg.barf_iter"]))
    net = hold_amd.build_from_scene(sc, sd_np, device="cuda:0")
    for node in net.nodes.values():
        node.params.defrost()  # HOLD.__init__ (hold.py:47-51)
        node.implicit_network.embedder_obj.step()  # as the recording script: BARF weights from the loaded counter
    net.train(train)
    return sc, net


def _z(g, prefix, nodes):
    z = z_in_reference_order({f"{n}.z_vals": g[f"{prefix}{n}.z_vals"] for n in nodes}, nodes)
    return {n: v.cuda() for n, v in z.items()}


def _batch(sc, frames, W, seed):
    b = syn.make_batch(sc, frames, syn.make_uv(W, W), W, W, seed=seed)
    return {k: torch.from_numpy(v).cuda() for k, v in b.items()}


def test_three_training_steps_of_the_reference_module_replayed_on_the_hip_path(gold_dir):
    from hold_amd.loss import Loss
    from hold_amd.optim import FlatAdam
    from hold_amd.train import training_step
    g = dict(np.load(os.path.join(gold_dir, "hold_steps.npz")))
    sc, net = _net(g, True)
    nodes = list(sc["entities"])
    opt = FlatAdam(net, lr=float(g["cfg.lr"]), clip_norm=float(g["cfg.clip"]))
    # configure_optimizers (hold.py:79-101): the same groups -- one 0.1 x lr group per node, then everything else at lr
    groups = [(gr["lr"], sum(p.numel() for p in gr["params"])) for gr in opt.reference_groups()]
    assert np.allclose(np.asarray(groups), g["cfg.groups"])
    pn = dict(net.named_parameters())
    recorded = [k[2:] for k in g if k.startswith("p.")]
    # every parameter the reference's three steps MOVED exists here under the same name and is trainable, and nothing else is
    # (the MANO layer's own pose / shape parameters are trainable by flag in the reference but never reached by a gradient)
    moved = sorted(n for n in recorded if float(g["dnorm." + n]) > 0.0)
    names = sorted(n for n, p in net.named_parameters() if p.requires_grad and p.numel() and n in recorded)
    assert set(moved) <= set(names) and len(moved) >= 100
    assert all(float(g["dnorm." + n]) == 0.0 for n in recorded if n not in names)
    assert {n for n, p in net.named_parameters() if p.requires_grad and p.numel()} <= set(recorded)
    p0 = {n: pn[n].detach().clone() for n in names}
    loss_fn = Loss()
    W, epoch = int(g["cfg.W"]), int(g["cfg.epoch"])
    for k in range(int(g["cfg.steps"])):
        step = int(g["cfg.first_step"]) + k
        pre = f"s{k}."
        batch = _batch(sc, g["cfg.frames"][k].tolist(), W, 1 + k)
        n_rand = len([1 for kk in g if kk.startswith(pre + "rand.")])
        assert n_rand == 2 * len(nodes) + 1  # (stratified z, final u) per node, then the background's stratified draw
        rng = {"bg_t": torch.from_numpy(g[pre + f"rand.{n_rand - 1}"]).cuda()}
        opt.zero_grad()
        loss, ld, out = training_step(net, loss_fn, batch, epoch, step, rng=rng, z_override=_z(g, pre, nodes))
        loss.backward()
        assert float(loss) == pytest.approx(float(g[pre + "loss"]), rel=1e-5), k
        for t in ("loss/rgb", "loss/sem"):
            assert float(ld[t]) == pytest.approx(float(g[pre + t]), rel=1e-5), (k, t)
        assert float(ld["loss/mano_cano"]) == 0.0 and float(ld["loss/opacity_sparse"]) == 0.0  # no canonical mesh before step 200
        for key in ("rgb", "semantics", "fg_rgb", "depth", "mask_prob", "right.mask_prob", "object.mask_prob", "bg_z_vals"):
            ref = g[pre + "out." + key]
            assert np.abs(out[key].detach().cpu().numpy().reshape(ref.shape) - ref).max() < 1e-4, (k, key)
        en = np.abs(out["normal"].detach().cpu().numpy() - g[pre + "out.normal"])
        assert np.quantile(en, 0.99) < 1e-4 and en.max() < 1e-3, (k, en.max())
        assert opt.grad_norm() == pytest.approx(float(g[pre + "grad_norm"]), rel=2e-4), k  # the norm Lightning's clip sees
        opt.step()
        assert int(net.nodes["object"].implicit_network.embedder_obj.alpha_iter) == int(g[pre + "barf_iter_after"])
    assert opt.step_count == int(g["cfg.steps"])
    worst_p, worst_d = ("", 0.0), ("", 0.0)
    for n in names:
        ph = _sample(pn[n]).cpu().double().numpy()
        dh = _sample(pn[n].detach() - p0[n]).cpu().double().numpy()
        # a strided sample of a large tensor is judged against the WHOLE tensor's norm, scaled to the sample's share
        share = np.sqrt(len(ph) / pn[n].numel())
        rp = np.linalg.norm(ph - g["p." + n]) / (float(g["pnorm." + n]) * share + 1e-30)
        rd = np.linalg.norm(dh - g["d." + n]) / (float(g["dnorm." + n]) * share + 1e-30) if float(g["dnorm." + n]) > 0 else 0.0
        worst_p = max(worst_p, (n, rp), key=lambda t: t[1])
        worst_d = max(worst_d, (n, rd), key=lambda t: t[1])
    print(f"three reference steps replayed: worst parameter error {worst_p}, worst update error {worst_d}")
    assert worst_p[1] < 1e-4, worst_p
    assert worst_d[1] < 5e-3, worst_d  # Adam divides by sqrt(v): elements with gradients below eps move by their last bits


def _replay_updates(g, sc, net):
    """bring a fresh net to the state the reference's frame was rendered in: the three recorded steps"""
    from hold_amd.loss import Loss
    from hold_amd.optim import FlatAdam
    from hold_amd.train import training_step
    nodes = list(sc["entities"])
    opt = FlatAdam(net, lr=float(g["cfg.lr"]), clip_norm=float(g["cfg.clip"]))
    net.train()
    for k in range(int(g["cfg.steps"])):
        pre = f"s{k}."
        batch = _batch(sc, g["cfg.frames"][k].tolist(), int(g["cfg.W"]), 1 + k)
        rng = {"bg_t": torch.from_numpy(g[pre + f"rand.{2 * len(nodes)}"]).cuda()}
        opt.zero_grad()
        loss, _, _ = training_step(net, Loss(), batch, int(g["cfg.epoch"]), int(g["cfg.first_step"]) + k, rng=rng,
                                   z_override=_z(g, pre, nodes))
        loss.backward()
        opt.step()


def _frame_batch(g, sc):
    W = int(g["cfg.inf_W"])
    b = syn.make_batch(sc, [int(g["cfg.inf_frame"])], syn.make_uv(W, W), W, W)
    batch = {k: torch.from_numpy(v) for k, v in b.items()}
    batch["total_pixels"] = torch.tensor([W * W])
    batch["pixel_per_batch"] = int(g["cfg.pixel_per_batch"])
    batch["img_size"] = [torch.tensor([W]), torch.tensor([W])]
    return batch


def _psnr(a, b):
    return 10 * np.log10(1.0 / max(float(((a - b) ** 2).mean()), 1e-20))


def test_inference_step_of_the_reference_module_replayed_on_the_hip_path(gold_dir):
    """HOLD.inference_step (hold.py:169-208: model.eval() -- the BARF masks stay ON, it is render.py:43-47 that switches them
    off -- pose-table rows, 512-pixel chunks, the merged vis keys + the batch) on a 64 x 64 frame.
    (a) BEFORE the first update -- identical weights on both sides -- with the reference's z_vals: every key at 1e-4 (rendered
        normals: 99 % at 1e-4, all at 1e-3; instance_map, an argmax, equal on 99.9 % of the pixels);
    (b) same state, the HIP sampler in the loop, in the reference's chunks and frame-at-once (hold_amd's default): rgb PSNR > 50 dB;
    (c) AFTER the three replayed updates (the two trajectories differ by ~1e-5 of a parameter's norm by then): rgb PSNR > 50 dB
        against the frame the reference rendered after ITS three updates, instance_map equal on 99.5 % of the pixels."""
    from hold_amd.train import inference_step
    g = dict(np.load(os.path.join(gold_dir, "hold_steps.npz")))
    sc, net = _net(g, True)
    nodes = list(sc["entities"])
    batch = _frame_batch(g, sc)
    epoch, step0, ppb = int(g["cfg.epoch"]), int(g["cfg.first_step"]), int(g["cfg.pixel_per_batch"])
    zo = z_in_reference_order({f"{n}.z_vals": g[f"inf0.{n}.z_vals"] for n in nodes}, nodes)
    out = inference_step(net, batch, epoch, step0, chunk_rays=ppb, z_override=zo)
    keys = [str(k) for k in g["inf0.keys"]]
    assert sorted(k for k in out.keys() if k in ("rgb", "instance_map", "bg_rgb_only") or "fg_rgb.vis" in k or "mask_prob" in k
                  or "normal" in k) == keys  # the key set hold.py:192-201 keeps
    assert not net.training
    for k in keys:
        ref, got = g["inf0.out." + k], out[k].numpy()
        assert got.shape == ref.shape, k
        if k == "instance_map":
            assert (got == ref).mean() > 0.999
        elif "normal" in k:
            en = np.abs(got - ref)
            assert np.quantile(en, 0.99) < 1e-4 and en.max() < 1e-3, (k, en.max())
        else:
            assert np.abs(got - ref).max() < 1e-4, (k, np.abs(got - ref).max())
    assert all(k in out for k in ("uv", "idx", "gt.rgb", "current_epoch", "global_step"))  # output.update(batch), hold.py:207
    # (b)
    out2 = inference_step(net, batch, epoch, step0, chunk_rays=ppb)
    assert _psnr(out2["rgb"].numpy(), g["inf0.out.rgb"]) > 50
    out3 = inference_step(net, batch, epoch, step0)
    assert _psnr(out3["rgb"].numpy(), g["inf0.out.rgb"]) > 50
    assert (out3["instance_map"].numpy() == g["inf0.out.instance_map"]).mean() > 0.995
    # (c)
    _replay_updates(g, sc, net)
    out4 = inference_step(net, batch, epoch, step0 + int(g["cfg.steps"]), chunk_rays=ppb)
    p_after, p_stale = _psnr(out4["rgb"].numpy(), g["inf.out.rgb"]), _psnr(out3["rgb"].numpy(), g["inf.out.rgb"])
    print(f"frame after three updates: PSNR {p_after:.1f} dB against the reference's (the un-updated frame: {p_stale:.1f} dB)")
    assert p_after > 50 and p_after > p_stale  # ... and it IS the updated model that matches best
    assert (out4["instance_map"].numpy() == g["inf.out.instance_map"]).mean() > 0.995


def test_training_sampler_with_the_reference_modules_draws(gold_dir):
    """the HIP sampler on the first recorded step, with the draws the reference's sampler took from torch's generator
    (stratified z, final u, the extras' permutation): same z_vals up to the discontinuous inverse-CDF stage"""
    g = dict(np.load(os.path.join(gold_dir, "hold_steps.npz")))
    sc, net = _net(g, True)
    nodes = list(sc["entities"])
    from hold_amd.train import with_params
    batch = with_params(net, _batch(sc, g["cfg.frames"][0].tolist(), int(g["cfg.W"]), 1), int(g["cfg.epoch"]), int(g["cfg.first_step"]))
    rng = {"bg_t": torch.from_numpy(g[f"s0.rand.{2 * len(nodes)}"]).cuda()}
    for i, n in enumerate(nodes):
        rng[n] = {"t_uniform": torch.from_numpy(g[f"s0.rand.{2 * i}"]).cuda(), "u_final": torch.from_numpy(g[f"s0.rand.{2 * i + 1}"]).cuda(),
                  "perm": torch.from_numpy(g[f"s0.perm.{i}"])}
    out = net(batch, rng=rng)
    for n in nodes:
        z, zr = out[n + ".z_vals"].cpu().numpy(), g[f"s0.{n}.z_vals"]
        assert z.shape == zr.shape
        dz = np.abs(z - zr)
        assert (dz > 1e-3).mean() < 0.02, (n, dz.max())
    assert np.abs(out["bg_z_vals"].cpu().numpy() - g["s0.out.bg_z_vals"]).max() < 1e-6
    mse = float(((out["rgb"].detach().cpu().numpy() - g["s0.out.rgb"]) ** 2).mean())
    assert 10 * np.log10(1.0 / max(mse, 1e-20)) > 50
