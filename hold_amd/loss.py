"""``Loss`` of the reference (code/src/hold/loss.py:9-93, terms in loss_terms.py:14-111) on the HIP kernels.

Same call ``Loss(args)(batch, model_outputs) -> dict`` with keys ``loss/rgb, loss/sem, [loss/eikonal,]
loss/mano_cano, loss/opacity_sparse, loss``.  The ray-wise terms (RGB L1, semantic L2, opacity sparsity of every node)
are ONE forward launch (``hold_pixel_loss_fwd``) and one backward launch instead of ~40 elementwise / indexing kernels;
the eikonal and MANO-canonical terms act on [B, 307] sample tensors and stay as torch expressions.
Two host synchronisations of the reference are removed without changing values: the ``if eikonal_loss > low_bnd``
branch (loss.py:86-88) becomes a device-side select, and the image-size probe (PIL, :28-31, unused afterwards) is
dropped.  ``image_scores`` is all ones in the reference (:24) and is folded away.
"""
from __future__ import annotations

import ctypes as C

import torch
from torch import nn

from . import _lib
from ._lib import call, ptr

_WS = {}


def reduce_workspace(dev):
    k = str(dev)
    if k not in _WS:
        _WS[k] = torch.empty(int(_lib.lib().hold_reduce_workspace_floats()), device=dev)
    return _WS[k]


class _PixelLossFn(torch.autograd.Function):
    """sums [10] of hold_pixel_loss_fwd (include/hold_hip.h) with the analytic backward."""

    @staticmethod
    def forward(ctx, rgb, gt_rgb, sem, gt_mask, offs, *mask_probs):
        dev = rgb.device
        n = len(mask_probs)
        rgb, sem = rgb.contiguous(), sem.contiguous()
        gt_rgb = gt_rgb.contiguous().float()
        gt_mask = gt_mask.contiguous().float()
        mps = [m.contiguous() for m in mask_probs]
        offs = [None if o is None else o.contiguous() for o in offs]
        nd = _lib.LossNodes()
        for i in range(n):
            nd.mask_prob[i] = mps[i].data_ptr()
            nd.off[i] = None if offs[i] is None else offs[i].data_ptr()
        sums = torch.empty(10, device=dev)
        call("hold_pixel_loss_fwd", ptr(rgb), ptr(gt_rgb), ptr(sem), ptr(gt_mask), rgb.shape[0], n, C.byref(nd), ptr(sums),
             ptr(reduce_workspace(dev)))
        ctx.save_for_backward(rgb, gt_rgb, sem, gt_mask, *mps)
        ctx.offs, ctx.n = offs, n
        return sums

    @staticmethod
    def backward(ctx, g):
        rgb, gt_rgb, sem, gt_mask, *mps = ctx.saved_tensors
        n, offs = ctx.n, ctx.offs
        d_rgb, d_sem = torch.empty_like(rgb), torch.empty_like(sem)
        d_mask = [torch.empty_like(m) if offs[i] is not None else None for i, m in enumerate(mps)]
        nd = _lib.LossNodes()
        for i in range(n):
            nd.mask_prob[i] = mps[i].data_ptr()
            nd.off[i] = None if offs[i] is None else offs[i].data_ptr()
            nd.d_mask[i] = None if d_mask[i] is None else d_mask[i].data_ptr()
        call("hold_pixel_loss_bwd", ptr(rgb), ptr(gt_rgb), ptr(sem), ptr(gt_mask), rgb.shape[0], n, C.byref(nd),
             ptr(g.contiguous().float()), ptr(d_rgb), ptr(d_sem))
        return (d_rgb, None, d_sem, None, None, *d_mask)


def get_eikonal_loss(grad_theta):
    return ((grad_theta.norm(2, dim=-1) - 1) ** 2).mean()


def get_mano_cano_loss(pred_sdf, gt_sdf, limit):
    return (torch.clamp(pred_sdf, -limit, limit) - torch.clamp(gt_sdf, -limit, limit)).abs().mean()


class Loss(nn.Module):
    def __init__(self, args=None):
        super().__init__()
        self.args = args
        self.milestone = 30000
        # ray-chunked steps (hold_amd.train.train_step): off-surface ray counts per node, accumulated on the device over
        # the chunks of the running step / the completed previous step
        self._off_acc, self._off_prev = {}, {}
        # ray tiles of ONE frame spread over ranks (bench.py --split rays): set to a process group (or True for the default
        # group) and the previous step's off-surface counts are summed over the ranks -- one all-reduce of a scalar per node
        # and step -- so that every rank normalises by the frame's exact count; `count_reduce` is the hook it goes through
        self.sync_group = None
        self.count_reduce = None

    def _chunked_sparse_den(self, nid, cnt, first_chunk, chunk_scale, own_scale, covers_all):
        """Denominator of a node's opacity-sparsity mean in a ray-chunked step.  The reference takes the mean over the
        off-surface rays of the WHOLE batch (loss_terms.get_opacity_sparse_loss); a chunk only knows its own count, and
        the later chunks' counts do not exist yet when this chunk is backpropagated.  Chunks are therefore normalised
        by the previous step's count of the rays THIS PROCESS owns, scaled to the whole batch by `own_scale` = batch rays /
        owned rays (1 unless the batch is a frame whose ray tiles are spread over ranks, bench.py --split rays: then every
        rank estimates the whole-frame count -- or, with `sync_group` / `count_reduce` set, knows it exactly from one scalar
        all-reduce per node and step -- and the ranks' terms ADD UP to one frame mean (round-3 advisor), in the
        first step by the chunk's own count scaled to the batch by `chunk_scale` -- so that the chunk terms add up to one
        batch mean instead of to `n_chunks` means.  A chunk that IS the whole batch uses its exact count.  Everything stays
        on the device (no host sync)."""
        cnt = cnt.detach()
        if first_chunk:
            if nid in self._off_acc:
                prev = self._off_acc[nid]
                red = self.count_reduce or (self._allreduce_count if self.sync_group is not None else None)
                if red is not None and own_scale != 1.0:
                    prev = red(prev)  # the frame's exact count of the previous step
                    self._off_total = getattr(self, "_off_total", set()) | {nid}
                self._off_prev[nid] = prev
            self._off_acc[nid] = cnt.clone()
        else:
            self._off_acc[nid] = self._off_acc.get(nid, torch.zeros_like(cnt)) + cnt
        if covers_all:
            return cnt
        prev = self._off_prev.get(nid)
        est = cnt * chunk_scale
        if prev is None:
            return est
        if nid in getattr(self, "_off_total", ()):
            own_scale = 1.0
        return torch.where(prev > 0, prev * own_scale, est)

    def _allreduce_count(self, t):
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            t = t.clone()
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=None if self.sync_group is True else self.sync_group)
        return t

    def forward(self, batch, model_outputs):
        rgb = model_outputs["rgb"]
        N = rgb.shape[0]
        rgb_gt = batch["gt.rgb"].reshape(-1, 3).to(rgb.device)
        mask_gt = batch["gt.mask"].reshape(-1).to(rgb.device)
        node_ids = [k.split(".")[0] for k in model_outputs.keys() if k.endswith(".index_off_surface")]
        offs = [model_outputs[f"{i}.index_off_surface"] for i in node_ids]
        mps = [model_outputs[f"{i}.mask_prob"].reshape(-1) for i in node_ids]
        sums = _PixelLossFn.apply(rgb, rgb_gt, model_outputs["semantics"], mask_gt, offs, *mps)
        if "hold_amd.n_total" in batch:  # ray-chunked step: normalise by the whole batch so chunk losses add up
            n_total = float(batch["hold_amd.n_total"])
            rgb_loss = sums[0] / (n_total + 1e-6)
        else:
            n_total = float(N)
            rgb_loss = sums[0] / (sums[2] + 1e-6)  # valid_pix[nan_filter].sum() (loss.py:38-44)
        sem_loss = sums[1] / n_total
        opacity_sparse_loss = 0.0
        chunked = "hold_amd.n_total" in batch
        # rays this process owns in the step (train_step): the whole batch, or its ray tile of a frame shared between ranks
        rays_owned = float(batch.get("hold_amd.rays_owned", n_total))
        for i, nid in enumerate(node_ids):
            num, cnt = sums[3 + 2 * i], sums[4 + 2 * i]
            if chunked:
                den = self._chunked_sparse_den(nid, cnt, bool(batch.get("hold_amd.frame_terms", True)), n_total / float(N),
                                               n_total / rays_owned, float(N) == n_total)
                # a chunk without off-surface rays contributes 0 (0 / 0 would poison the accumulated gradient bucket)
                opacity_sparse_loss = opacity_sparse_loss + torch.where(den > 0, num / den.clamp_min(1.0), torch.zeros_like(num))
            else:
                opacity_sparse_loss = opacity_sparse_loss + num / cnt
        eikonal_loss = 0.0
        for k in model_outputs.keys():
            if "grad_theta" in k:
                eikonal_loss = eikonal_loss + get_eikonal_loss(model_outputs[k])
        mano_cano_loss = 0.0
        for k in model_outputs.keys():
            if "pts2mano_sdf_cano" in k:
                nid = k.split(".")[0]
                mano_cano_loss = mano_cano_loss + get_mano_cano_loss(model_outputs[f"{nid}.pred_sdf"],
                                                                     model_outputs[k].detach(), 0.01)
        # per-frame terms (evaluated once per process, with its first chunk): with the frame's ray tiles spread over ranks
        # whose gradients are SUMMED, each rank contributes its share of the frame -- not the whole term once per rank
        share = rays_owned / n_total if chunked else 1.0
        progress = min(self.milestone, int(model_outputs["step"]))
        w_sem = 1.1 + (0.1 - 1.1) * progress / self.milestone  # torch.linspace(1.1, 0.1, milestone + 1)[progress]
        w_sparse = progress / self.milestone
        loss_dict = {"loss/rgb": rgb_loss * 1.0, "loss/sem": sem_loss * w_sem}
        eikonal_loss = eikonal_loss * 0.00001
        if torch.is_tensor(eikonal_loss):  # loss.py:86-88 without the host sync (the threshold acts on the whole term)
            loss_dict["loss/eikonal"] = torch.where(eikonal_loss > 0.0008, eikonal_loss, torch.zeros_like(eikonal_loss)) * share
        loss_dict["loss/mano_cano"] = mano_cano_loss * (5.0 * share)
        loss_dict["loss/opacity_sparse"] = opacity_sparse_loss * w_sparse
        loss_dict["loss"] = sum(loss_dict[k] for k in list(loss_dict.keys()))
        return loss_dict
