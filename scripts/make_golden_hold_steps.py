"""Generate tests/golden/hold_steps.npz by driving the REFERENCE's own Lightning module on CPU (build container only;
/root/reference does not travel to the GPU box in any form, so what travels is this fixture):

  src.hold.hold.HOLD(opt, args)                      code/src/hold/hold.py:26-55 (the reference's HOLDNet inside)
  .configure_optimizers()  -> torch.optim.Adam      hold.py:79-101 (0.1 x lr group per node + the main group, eps 1e-8)
  .training_step(batch) x 3                          hold.py:110-137 (wubba_lubba_dub_dub, pose-table rows, HOLDNet.forward in
                                                     training mode, the reference's Loss)
  loss.backward(); clip_grad_norm_(0.5); Adam.step   code/train.py:28-73 (Trainer(gradient_clip_val=0.5))
  .inference_step(batch) on a 64 x 64 frame          hold.py:169-208 (split_input in 512-pixel chunks, merge_output)

with the synthetic scene and weights of the parity tests (hold_amd.synthetic, seed 1).  Recorded per training step: the loss
terms, the total gradient norm before the clip, every node's z_vals, the draws the step took from torch's generator
(torch.rand / torch.randperm, in call order), a few outputs; after the third step: every parameter (tensors above 4 096
elements as a 1 024-element stride sample + their norm) and its three-step update; from inference_step: the merged vis keys and
the z_vals the reference's sampler produced chunk by chunk.  tests/test_dropin_gpu.py replays the same three steps and the
same frame through hold_amd's HIP path on the MI355X and holds them to these numbers.

    python scripts/make_golden_hold_steps.py
"""
import os
import sys
import tempfile

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from hold_amd import synthetic as syn  # noqa: E402
from oracle import ref_shim  # noqa: E402
from make_golden import RandRecorder, np_  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
CFG = dict(n_frames=4, W=6, frames=[[0, 2], [1, 3], [2, 0]], steps=3, first_step=1, epoch=0, lr=5e-4, clip=0.5, barf_iter=3999,
           inf_W=64, inf_frame=1, pixel_per_batch=512)


def sample(t):
    t = t.detach().reshape(-1)
    return t if t.numel() <= 4096 else t[:: max(1, t.numel() // 1024)][:1024]


def ref_batch(b, png):
    """a flattened hold_amd.synthetic batch in the shape the reference's DataLoader hands to training_step: a leading batch
    dimension of 1 over [n_images, ...] (hold_utils.wubba_lubba_dub_dub folds the two), idx as a list of per-image tensors"""
    out = {k: torch.from_numpy(v)[None] for k, v in b.items() if k != "idx"}
    out["idx"] = [torch.tensor([int(i)]) for i in b["idx"]]
    out["im_path"] = [[png]]
    return out


def main():
    from PIL import Image
    os.makedirs(GOLD, exist_ok=True)
    torch.set_num_threads(os.cpu_count())
    ref_shim.install()
    import src.hold.hold as H
    from common.xdict import xdict
    sc = syn.make_scene(n_frames=CFG["n_frames"])
    wd = ref_shim.prepare_workdir(sc)
    opt = ref_shim.load_opt()
    opt.model.scene_bounding_sphere = sc["scene_bounding_sphere"]
    args = ref_shim.make_args(n_images=sc["n_frames"], lr=CFG["lr"])
    torch.manual_seed(1)
    np.random.seed(1)
    with ref_shim.chdir(wd):
        hold = H.HOLD(opt, args)
    assert type(hold.model).__module__ == "src.hold.hold_net"  # the reference's own model, not ours
    sd_np = syn.make_state_dict(sc, barf_iter=CFG["barf_iter"])
    hold.model.load_state_dict({k: torch.as_tensor(v) for k, v in sd_np.items()}, strict=False)
    for node in hold.model.nodes.values():
        node.implicit_network.embedder_obj.step()  # refresh the BARF weights from the loaded counter (as tests/parity_common.hip_net)
    hold.log = lambda *a, **k: None  # Lightning's logger call (the shim's LightningModule is a plain nn.Module)
    hold.current_epoch = CFG["epoch"]
    optim = hold.configure_optimizers()[0][0]
    png = os.path.join(tempfile.mkdtemp(prefix="hold_png_"), "im.png")
    Image.fromarray(np.zeros((CFG["W"], CFG["W"], 3), np.uint8)).save(png)
    names = [n for n, p in hold.model.named_parameters() if p.requires_grad and p.numel()]
    pn = dict(hold.model.named_parameters())
    p0 = {n: pn[n].detach().clone() for n in names}
    gold = {"cfg.frames": np.asarray(CFG["frames"]), "cfg.W": CFG["W"], "cfg.steps": CFG["steps"], "cfg.first_step": CFG["first_step"],
            "cfg.epoch": CFG["epoch"], "cfg.lr": CFG["lr"], "cfg.clip": CFG["clip"], "cfg.barf_iter": CFG["barf_iter"],
            "cfg.inf_W": CFG["inf_W"], "cfg.inf_frame": CFG["inf_frame"], "cfg.pixel_per_batch": CFG["pixel_per_batch"],
            "cfg.groups": np.asarray([(g["lr"], sum(p.numel() for p in g["params"])) for g in optim.param_groups])}
    uv = syn.make_uv(CFG["W"], CFG["W"])

    # z_vals of every sampler call, in call order (training: one call per node and step; inference: one per node and chunk)
    z_log = []
    for nid, node in hold.model.nodes.items():
        orig = node.ray_sampler.get_z_vals

        def rec(*a, _orig=orig, _nid=nid, **k):
            z = _orig(*a, **k)
            z_log.append((_nid, z.detach().clone()))
            return z

        node.ray_sampler.get_z_vals = rec

    def inference(tag, step, with_z):
        """HOLD.inference_step on the 64 x 64 frame; with_z: also the z_vals its sampler produced, chunk by chunk"""
        W = CFG["inf_W"]
        b = syn.make_batch(sc, [CFG["inf_frame"]], syn.make_uv(W, W), W, W)
        batch = {k: torch.from_numpy(v) for k, v in b.items()}
        batch["total_pixels"] = torch.tensor([W * W])
        batch["pixel_per_batch"] = CFG["pixel_per_batch"]
        batch["img_size"] = [torch.tensor([W]), torch.tensor([W])]
        xdict.to = lambda self, dev: self  # hold.py:170 moves the batch to "cuda"
        del z_log[:]
        hold.global_step = step
        with torch.no_grad(), ref_shim.chdir(wd):
            out = hold.inference_step(batch)
        n_chunks = (W * W + CFG["pixel_per_batch"] - 1) // CFG["pixel_per_batch"]
        assert len(z_log) == n_chunks * len(hold.model.nodes)
        if with_z:
            for nid in hold.model.nodes:
                gold[f"{tag}.{nid}.z_vals"] = np_(torch.cat([z for n, z in z_log if n == nid], 0))
        keys = [k for k in out.keys() if k in ("rgb", "instance_map", "bg_rgb_only") or "fg_rgb.vis" in k or "mask_prob" in k or "normal" in k]
        for k in keys:
            gold[f"{tag}.out." + k] = np_(out[k])
        gold[f"{tag}.keys"] = np.asarray(sorted(keys))
        print(tag, "inference_step keys:", sorted(keys))

    # the frame BEFORE the first update (identical weights on both sides of the comparison: every key is held to 1e-4) ...
    inference("inf0", CFG["first_step"], True)
    hold.train()
    for k in range(CFG["steps"]):
        hold.global_step = CFG["first_step"] + k
        b = syn.make_batch(sc, CFG["frames"][k], uv, CFG["W"], CFG["W"], seed=1 + k)
        del z_log[:]
        torch.manual_seed(100 + k)
        with RandRecorder() as rr, ref_shim.chdir(wd):
            captured = {}
            fwd = hold.model.forward

            def model_fwd(inp, _fwd=fwd, _c=captured):
                o = _fwd(inp)
                _c["out"] = o
                return o

            hold.model.forward = model_fwd
            try:
                loss = hold.training_step(ref_batch(b, png))
            finally:
                hold.model.forward = fwd
            ld = hold.loss(xdict({"idx": torch.from_numpy(b["idx"]), "im_path": [[png]], "gt.rgb": torch.from_numpy(b["gt.rgb"]),
                                  "gt.mask": torch.from_numpy(b["gt.mask"])}), captured["out"])  # the terms, for the record
        assert float(ld["loss"]) == float(loss)
        optim.zero_grad()
        loss.backward()
        gn = torch.nn.utils.clip_grad_norm_(hold.parameters(), CFG["clip"])  # Trainer(gradient_clip_val=0.5), train.py:30
        optim.step()
        pre = f"s{k}."
        gold[pre + "loss"] = np_(loss)
        for kk, v in ld.items():
            gold[pre + kk] = np_(v) if torch.is_tensor(v) else np.float32(v)
        gold[pre + "grad_norm"] = np_(gn)
        out = captured["out"]
        for kk in ("rgb", "semantics", "fg_rgb", "normal", "depth", "mask_prob", "right.mask_prob", "object.mask_prob", "bg_z_vals"):
            gold[pre + "out." + kk] = np_(out[kk])
        assert [n for n, _ in z_log] == list(hold.model.nodes.keys())
        for nid, z in z_log:
            gold[pre + nid + ".z_vals"] = np_(z)
        for i, r in enumerate(rr.rand):
            gold[pre + f"rand.{i}"] = np_(r)
        for i, r in enumerate(rr.perm):
            gold[pre + f"perm.{i}"] = np_(r)
        gold[pre + "barf_iter_after"] = int(hold.model.nodes["object"].implicit_network.embedder_obj.alpha_iter)
        print(f"step {hold.global_step}: loss {float(loss):.6f} " + " ".join(f"{kk}={float(v):.5f}" for kk, v in ld.items() if kk != "loss")
              + f" |grad| {float(gn):.4f}  draws: {len(rr.rand)} rand, {len(rr.perm)} perm")

    for n in names:
        p, d = pn[n].detach(), pn[n].detach() - p0[n]
        gold["p." + n] = np_(sample(p))
        gold["pnorm." + n] = np_(p.norm())
        gold["d." + n] = np_(sample(d))
        gold["dnorm." + n] = np_(d.norm())

    # ... and AFTER the three updates (outputs only: the two trajectories differ by ~1e-5 of a parameter's norm by then, which a
    # density of 1 / beta = 10 per unit of sdf turns into up to 1e-3 of a colour; held to a PSNR)
    inference("inf", CFG["first_step"] + CFG["steps"], False)
    np.savez_compressed(os.path.join(GOLD, "hold_steps.npz"), **gold)
    print("wrote", os.path.join(GOLD, "hold_steps.npz"), os.path.getsize(os.path.join(GOLD, "hold_steps.npz")) // 1024, "KiB")


if __name__ == "__main__":
    main()
