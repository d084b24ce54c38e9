"""Training-step driver around HOLDNet: the per-pixel losses active from step 0 of the reference
(code/src/hold/loss.py:17-93 with loss_terms.get_rgb_loss / get_sem_loss; the eikonal / MANO-cano /
opacity-sparse terms need kaolin-derived targets and switch on only after the first canonical-mesh
spawn -- SURVEY.md 8(f-2)), ray-chunked gradient accumulation, and the data-parallel gradient
all-reduce (hold_amd.parallel)."""
from __future__ import annotations

import torch

SEGM = (25, 100, 200)  # class boundaries of loss_terms.get_sem_loss


def w_sem(step, milestone=30000):
    p = min(milestone, int(step))
    return 1.1 + (0.1 - 1.1) * p / milestone  # torch.linspace(1.1, 0.1, milestone+1)[progress]


def pixel_losses(out, gt_rgb, gt_mask, n_total, step=0):
    """sum-reduced over the given rays and divided by n_total (= valid_pix.sum() of the whole batch) so
    chunk losses add up to the reference's batch loss."""
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


def train_step(net, inp, chunk_rays, step=0, epoch=0):
    """fwd + loss + bwd over all rays of `inp` (uv [B,P,2]) in chunks of chunk_rays per frame; gradients
    accumulate in .grad.  Returns (loss value, rays processed)."""
    B, P = inp["uv"].shape[:2]
    n_total = B * P
    total = 0.0
    for lo in range(0, P, chunk_rays):
        hi = min(P, lo + chunk_rays)
        c = chunked_input(inp, lo, hi)
        c["current_epoch"], c["global_step"] = epoch, step
        for node in net.nodes.values():  # pose-table lookups carry gradients: one graph per chunk
            c.update(node.params(c["idx"]))
        out = net(c)
        loss, _ = pixel_losses(out, c["gt.rgb"].reshape(-1, 3), c["gt.mask"].reshape(-1), n_total, step)
        loss.backward()
        total += float(loss.detach())
    return total, n_total


@torch.no_grad()
def render_frame(net, inp, chunk_rays, keys=("rgb", "normal", "mask_prob", "depth", "instance_map")):
    """inference_step-style chunked rendering (code/src/hold/hold.py:169-208) without the per-chunk D2H."""
    B, P = inp["uv"].shape[:2]
    outs = {k: [] for k in keys}
    for lo in range(0, P, chunk_rays):
        c = chunked_input(inp, lo, min(P, lo + chunk_rays))
        for node in net.nodes.values():
            c.update(node.params(c["idx"]))
        o = net(c)
        for k in keys:
            outs[k].append(o[k].reshape(B, -1, *o[k].shape[1:]))
    return {k: torch.cat(v, 1).reshape(B * P, *v[0].shape[2:]) for k, v in outs.items()}
