"""Free functions of the reference's engine layer that sit on the drop-in boundary, kernel-backed:
``sdf_func_with_deformer`` (code/src/engine/volsdf_utils.py:150-169) and ``compute_gradient_samples`` (:19-48).  The fused
node path in hold_net.py does not go through these (it launches the same kernels without materialising [P,257] outputs);
they exist so that reference-side callers keep working.  ``density2weight`` (:220-251) has NO standalone equivalent here:
its only caller in the reference is the volumetric renderer (hold_utils.py:245), which HOLDNet's compositor kernel
(hold_composite_fwd / _bwd: density, weights, transmittance and all renders of a ray in one launch) replaces as a whole --
an eager-torch restatement would be a CPU/eager path inside the product package (round 3 had one; removed)."""
from __future__ import annotations

import torch


def sdf_func_with_deformer(deformer, sdf_fn, training, x, deform_info):
    """x [P,3] deformed-space points of B frames -> (sdf [B,P/B,1], x_c [B,P/B,3], feature [B,P/B,256])."""
    cond, tfs = deform_info["cond"], deform_info["tfs"]
    verts = deform_info.get("verts")
    B = tfs.shape[0]
    x = x.reshape(B, -1, 3)
    # tfs: [B,4,4] for the object (object_node.py:76-79), [B,16,4,4] for a hand -- both deformers take the same call
    x_c, _ = deformer.forward(x, tfs, return_weights=False, inverse=True, verts=verts)
    out = sdf_fn(x_c, cond)
    return out[:, :, 0:1], x_c, out[:, :, 1:]


def compute_gradient_samples(pt_in_space_sampler, implicit_network, cond, num_pixels, verts_c, local_sigma=0.008,
                             global_ratio=0.20):
    """volsdf_utils.py:19-48 -> grad_theta [B, n, 3] with a second-order graph to the weights."""
    if verts_c is not None:
        idx = torch.randperm(verts_c.shape[1])[:num_pixels].to(verts_c.device)
        sample = pt_in_space_sampler.get_points(torch.index_select(verts_c, 1, idx), local_sigma=local_sigma,
                                                global_ratio=global_ratio)
    else:
        B, dev = cond["pose"].shape[0], cond["pose"].device
        sample = torch.rand(B, num_pixels, 3).to(dev) * 0.6 - 0.3
    B, n, _ = sample.shape
    return implicit_network.gradient(sample.reshape(-1, 3), cond).view(B, n, 3)
