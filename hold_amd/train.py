"""Step drivers around HOLDNet -- the logic of the reference's Lightning module (code/src/hold/hold.py) without
Lightning: ``training_step`` (:110-137), ``inference_step`` (:169-208) rendered frame-at-once with ONE device-to-host
copy (the reference renders 512-pixel chunks and copies every chunk to the CPU), and ``train_step`` =
ray-chunked forward + loss + backward with gradient accumulation for frames that do not fit one call (512x512 and up;
the reference only ever trains on 1 280-ray batches).  The optimiser step lives in hold_amd.optim.FlatAdam."""
from __future__ import annotations

import torch

from .loss import Loss
from .xdict import output_class

SEGM = (25, 100, 200)  # class boundaries of loss_terms.get_sem_loss


def w_sem(step, milestone=30000):
    p = min(milestone, int(step))
    return 1.1 + (0.1 - 1.1) * p / milestone  # torch.linspace(1.1, 0.1, milestone+1)[progress]


def pixel_losses(out, gt_rgb, gt_mask, n_total, step=0):
    """rgb + semantic terms only (the loss before the first canonical mesh exists), sum-reduced over the given rays and
    divided by n_total so chunk losses add up to the reference's batch loss.  Kept for the round-1 comparisons;
    the full loss is hold_amd.loss.Loss."""
    rgb = out["rgb"]
    nan_filter = ~torch.any(rgb.isnan(), dim=1)
    rgb_loss = (rgb[nan_filter] - gt_rgb[nan_filter]).abs().sum() / (n_total + 1e-6)
    cls = torch.zeros_like(gt_mask)
    cls[(gt_mask >= SEGM[0]) & (gt_mask < SEGM[1])] = 1
    cls[(gt_mask >= SEGM[1]) & (gt_mask < SEGM[2])] = 2
    cls[gt_mask >= SEGM[2]] = 3
    onehot = torch.nn.functional.one_hot(cls, 4).to(rgb.dtype)
    sem_loss = ((out["semantics"] - onehot) ** 2).sum() / n_total
    return rgb_loss + sem_loss * w_sem(step), dict(rgb=rgb_loss.detach(), sem=sem_loss.detach())


def chunked_input(inp, lo, hi):
    c = dict(inp)
    for k in ("uv", "gt.rgb", "gt.mask"):
        if k in c:
            c[k] = inp[k][:, lo:hi].contiguous()
    return c


def with_params(net, batch, epoch, step):
    """what training_step / inference_step add to the batch (hold.py:114-121,173-179): counters + pose-table rows."""
    b = dict(batch)
    b["current_epoch"], b["global_step"] = epoch, step
    for node in net.nodes.values():
        b.update(node.params(b["idx"]))
    return b


def training_step(net, loss_fn, batch, epoch=0, step=0, **forward_kw):
    """hold.py:110-137 on an already flattened batch (uv [B,P,2], gt.rgb [B,P,3], gt.mask [B,P], idx [B], ...):
    -> (loss tensor, loss dict, model outputs).  ``forward_kw``: HOLDNet.forward's test hooks (rng = recorded draws,
    z_override = recorded z_vals: tests/test_dropin_gpu.py replays the reference's own steps with them)."""
    b = with_params(net, batch, epoch, step)
    out = net(b, **forward_kw)
    ld = loss_fn(b, out)
    return ld["loss"], ld, out


def train_step(net, inp, chunk_rays, step=0, epoch=0, loss_fn=None, n_total=None, n_chunks=None):
    """fwd + loss + bwd over all rays of `inp` (uv [B,P,2]) in chunks of chunk_rays per frame; gradients accumulate in
    .grad.  Per-frame (not per-ray) loss terms -- eikonal, MANO-canonical -- are evaluated with the first chunk only;
    the BARF counter steps once per call (the reference steps it once per training_step).
    ``n_total``: the ray count the ray-wise loss terms are normalised by (default: the rays of ``inp``; a rank that owns a
    ray tile of a frame passes the frame's total so that the ranks' gradients ADD UP to the whole-frame gradient -- the
    opacity-sparsity denominator and the per-frame terms are scaled by the tile's share of the frame, hold_amd.loss.Loss).
    ``n_chunks``: run exactly this many forwards, over balanced chunks (parallel.chunk_bounds) -- ranks that own ray tiles of
    one frame pass parallel.tile_chunks(frame rays, world, chunk_rays) so that every rank issues the same sequence of
    collectives (sampler rounds, loss counts) even when the tiles' own ceil(P / chunk_rays) differ.
    Returns (loss -- a device scalar, read it once per step at most --, rays processed)."""
    B, P = inp["uv"].shape[:2]
    n_total = B * P if n_total is None else int(n_total)
    total = None
    auto = net.auto_step_embedding
    net.auto_step_embedding = False
    if n_chunks is None:
        bounds = [(lo, min(P, lo + chunk_rays)) for lo in range(0, P, chunk_rays)]
    else:
        from .parallel import chunk_bounds
        bounds = chunk_bounds(P, int(n_chunks))
        if any(hi - lo > chunk_rays for lo, hi in bounds) or any(hi == lo for lo, hi in bounds):
            raise ValueError(f"train_step: {n_chunks} chunks of {P} rays do not fit chunk_rays = {chunk_rays} (or leave a "
                             "chunk empty): pass parallel.tile_chunks(frame rays, world, chunk_rays)")
    try:
        for ci, (lo, hi) in enumerate(bounds):
            c = with_params(net, chunked_input(inp, lo, hi), epoch, step)  # pose-table lookups: one graph per chunk
            c["hold_amd.frame_terms"] = ci == 0
            c["hold_amd.n_total"] = n_total
            c["hold_amd.rays_owned"] = B * P
            out = net(c)
            if loss_fn is None:
                loss, _ = pixel_losses(out, c["gt.rgb"].reshape(-1, 3), c["gt.mask"].reshape(-1), n_total, step)
            else:
                loss = loss_fn(c, out)["loss"]
            loss.backward()
            total = loss.detach() if total is None else total + loss.detach()  # no host read per chunk
    finally:
        net.auto_step_embedding = auto
    if net.training and auto:
        net.step_embedding()
    return total, B * P


@torch.no_grad()
def render_frame(net, inp, chunk_rays, keys=("rgb", "normal", "mask_prob", "depth", "instance_map"), n_chunks=None):
    """chunked rendering that stays on the device (no per-chunk D2H); ``n_chunks`` as in train_step (ray tiles over ranks)."""
    B, P = inp["uv"].shape[:2]
    outs = {k: [] for k in keys}
    if n_chunks is None:
        bounds = [(lo, min(P, lo + chunk_rays)) for lo in range(0, P, chunk_rays)]
    else:
        from .parallel import chunk_bounds
        bounds = chunk_bounds(P, int(n_chunks))
        if any(hi - lo > chunk_rays or hi == lo for lo, hi in bounds):
            raise ValueError(f"render_frame: {n_chunks} chunks of {P} rays do not fit chunk_rays = {chunk_rays}")
    for lo, hi in bounds:
        c = chunked_input(inp, lo, hi)
        for node in net.nodes.values():
            c.update(node.params(c["idx"]))
        o = net(c)
        for k in keys:
            outs[k].append(o[k].reshape(B, -1, *o[k].shape[1:]))
    return {k: torch.cat(v, 1).reshape(B * P, *v[0].shape[2:]) for k, v in outs.items()}


VIS_KEYS = ("rgb", "instance_map", "bg_rgb_only")  # + every key containing fg_rgb.vis / mask_prob / normal (hold.py:193-199)


@torch.no_grad()
def inference_step(net, batch, epoch=0, step=0, chunk_rays=65536, no_vis=False, render_downsample=1, device="cuda",
                   z_override=None):
    """hold.py:169-208: eval-mode render of the batch's full pixel grid.  Same output mapping as the reference
    (merged vis keys + the batch itself), but the frame is rendered in chunks of ``chunk_rays`` (default 65 536 instead
    of the dataset's ``pixel_per_batch`` = 512) that stay on the device, and copied to the host once at the end.
    ``z_override`` (test hook): {node: z_vals [B * total_pixels, S]} used instead of the sampler's, chunk by chunk."""
    XD = output_class()
    to = lambda v: v.to(device) if torch.is_tensor(v) else v
    b = {k: to(v) for k, v in batch.items()}
    # model.eval() only (hold.py:171): nn.Module.eval() is train(False) on the children, it does NOT call the embedders' own
    # eval() -- the BARF masks stay on in a validation pass during training; render.py:43-47 switches them off itself before it
    # calls inference_step: disable_barf() below.  (Until round 6 this function did it too: found by replaying the reference
    # module's own inference_step, tests/test_dropin_gpu.py.)
    net.eval()
    b = with_params(net, b, epoch, step)
    output = {}
    if not no_vis:
        if render_downsample != 1:
            b = downsample_rendering(b, render_downsample)
        total = int(b["total_pixels"][0]) if "total_pixels" in b else b["uv"].shape[1]
        B = b["uv"].shape[0]
        parts = []
        for lo in range(0, total, chunk_rays):
            c = dict(b)
            c["uv"] = b["uv"][:, lo:lo + chunk_rays].contiguous()
            zo = None
            if z_override is not None:
                zo = {n: z.reshape(B, total, -1)[:, lo:lo + chunk_rays].reshape(-1, z.shape[-1]).contiguous().to(device)
                      for n, z in z_override.items()}
            o = net(c, z_override=zo)
            keep = {k: v for k, v in o.items()
                    if k in VIS_KEYS or "fg_rgb.vis" in k or "mask_prob" in k or "normal" in k}
            parts.append(keep)
        for k in parts[0]:
            v = [p[k] for p in parts]
            if v[0].dim() == 1:  # merge_output (datasets/utils.py:326-341)
                output[k] = torch.cat([t.reshape(B, -1, 1) for t in v], 1).reshape(B * total)
            else:
                output[k] = torch.cat([t.reshape(B, -1, t.shape[-1]) for t in v], 1).reshape(B * total, -1)
        output = {k: v.detach().cpu() for k, v in output.items()}  # the only D2H of the frame
    output.update({k: v for k, v in b.items() if k not in output})
    return XD(output)


def disable_barf(net):
    """render.py:43-47: `disable barf masks` -- the embedders of every node's implicit network and of the two background
    networks; sticky, as in the reference (BarfEmbedder.eval sets no_barf, embedders.py:124-125)."""
    for node in net.nodes.values():
        node.implicit_network.embedder_obj.eval()
    for name in ("bg_implicit_network", "bg_rendering_network"):
        emb = getattr(getattr(net.background, name, None), "embedder_obj", None)
        if emb is not None and hasattr(emb, "eval"):
            emb.eval()


def downsample_rendering(batch, k):
    """hold_utils.downsample_rendering (code/src/hold/hold_utils.py:306-331): keep every k-th pixel row / column."""
    im_h, im_w = int(batch["img_size"][0]), int(batch["img_size"][1])
    n = im_h * im_w
    out = dict(batch)
    nh, nw = im_h, im_w
    for key, val in batch.items():
        if torch.is_tensor(val) and val.dim() >= 2 and val.shape[1] == n:
            if val.dim() == 2:
                v = val.view(val.shape[0], im_h, im_w)[:, ::k, ::k]
                nh, nw = v.shape[1:]
                out[key] = v.reshape(val.shape[0], -1)
            else:
                v = val.view(val.shape[0], im_h, im_w, val.shape[-1])[:, ::k, ::k, :]
                nh, nw = v.shape[1:3]
                out[key] = v.reshape(val.shape[0], -1, val.shape[-1])
    dev = batch["uv"].device
    out["img_size"] = [torch.tensor([nh], device=dev), torch.tensor([nw], device=dev)]
    out["total_pixels"] = torch.tensor([nh * nw], device=dev)
    return out


__all__ = ["Loss", "training_step", "train_step", "inference_step", "render_frame", "pixel_losses", "disable_barf"]
