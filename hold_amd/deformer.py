"""Point deformers with the reference's call surface, on the HIP kernels.

``MANODeformer(max_dist, K, betas, is_rhand)`` / ``ObjectDeformer()`` are plain classes (not Modules) exactly as in
code/src/model/mano/deformer.py:6-143 and code/src/model/obj/deformer.py:5-46; module-level ``skinning`` mirrors
deformer.py:145-170.  Inverse / forward LBS of query points run in ``hold_knn_invlbs_fwd`` (KNN K=15 weight lookup
fused with the 4x4 blend + inverse), ``hold_invskin_fwd`` and ``hold_skin_fwd``; the outlier mask (unused downstream,
volsdf_utils.py:167-168) comes from ``hold_knn1_fwd``.  Weights are detached as in the reference (deformer.py:101);
gradients w.r.t. ``tfs`` flow through ``hold_invskin_bwd`` (used inside the fused node path, hold_net._FieldFn).
"""
from __future__ import annotations

import torch

from . import kernels as K
from ._lib import call, ptr


def _pts4(x):
    """[B,N,3] -> contiguous [B*N,4] (the kernels read 16-byte rows)."""
    B, N, _ = x.shape
    p = torch.zeros(B * N, 4, device=x.device, dtype=torch.float32)
    p[:, :3] = x.reshape(-1, 3)
    return p


def skinning(x, w, tfs, inverse=False):
    """x [B,N,3], w [B,N,J], tfs [B,J,4,4] -> [B,N,3]  (deformer.py:145-170)."""
    assert x.dim() == 3 and w.dim() == 3 and tfs.dim() == 4
    assert x.shape[0] == w.shape[0] == tfs.shape[0] and x.shape[1] == w.shape[1]
    B, N, _ = x.shape
    J = tfs.shape[1]
    out = torch.empty(B * N, 4, device=x.device)
    wf = w.detach().reshape(B * N, J).contiguous().float() if J > 1 else None
    t = tfs.detach().reshape(B, J, 16).contiguous().float()
    (K.invskin_fwd if inverse else K.skin_fwd)(_pts4(x), B * N, N, wf, t, J, out)
    return out[:, :3].reshape(B, N, 3)


def _min_dist(x, verts):
    """sqrt of the squared distance to the nearest vertex, clamped as the reference does (d2 <= 4): [B,N]."""
    B, N, _ = x.shape
    q = x.detach().contiguous().float()
    t = verts.detach().contiguous().float()
    d2 = torch.empty(B, N, device=x.device)
    idx = torch.empty(B, N, dtype=torch.int32, device=x.device)
    call("hold_knn1_fwd", ptr(q), B, N, ptr(t), t.shape[1], ptr(d2), ptr(idx))
    return torch.sqrt(torch.clamp(d2, max=4.0))


class KNNDeformer:
    def __init__(self, max_dist=0.1, K=15, betas=None, server=None):
        assert K == 15, "hold_knn_invlbs_fwd keeps a register top-15 (the only K HOLD uses, mano_node.py:28)"
        self.max_dist, self.K, self.server = max_dist, K, server

    # canonical vertices / skinning table: the server's canonical pose output (deformer.py:20-32)
    @property
    def verts(self):
        return self.server.verts_c

    @property
    def skin_weights(self):
        return self.server.human_layer.lbs_weights[None]

    def query_skinning_weights_multi(self, pts, verts, skin_weights=None):
        B, N, _ = pts.shape
        w = torch.empty(B * N, 16, device=pts.device)
        v = verts if verts.shape[0] == B else verts.expand(B, -1, -1)
        K.knn_invlbs(_pts4(pts), B * N, N, v.detach().contiguous().float(),
                     self.server.human_layer.lbs_weights.contiguous(), w_out=w)
        return w.view(B, N, 16), _min_dist(pts, v) > self.max_dist

    def forward(self, x, tfs, return_weights=True, inverse=False, verts=None):
        assert x.dim() == 3 and tfs.dim() == 4 and x.shape[0] == tfs.shape[0] and tfs.shape[2:] == (4, 4)
        if x.shape[0] == 0:
            return x
        v = self.verts if verts is None else verts
        weights, outlier = self.query_skinning_weights_multi(x, v)
        if return_weights:
            return weights
        return skinning(x, weights, tfs, inverse=inverse), outlier

    __call__ = forward

    def forward_skinning(self, xc, cond, tfs):
        weights, _ = self.query_skinning_weights_multi(xc, self.verts)
        return skinning(xc, weights, tfs, inverse=False)

    def query_weights(self, xc):
        return self.query_skinning_weights_multi(xc, self.verts)[0]


class MANODeformer(KNNDeformer):
    def __init__(self, max_dist, K, betas, is_rhand, server=None, mano_model=None):
        if server is None:  # the reference builds its own server here (deformer.py:127-142)
            from .mano import MANOServer
            server = MANOServer(betas=betas, is_rhand=is_rhand, model=mano_model)
        super().__init__(max_dist=max_dist, K=K, betas=betas, server=server)


class ObjectDeformer:
    def __init__(self):
        self.max_dist = 0.1

    def forward(self, x, tfs, return_weights=None, inverse=False, verts=None):
        assert x.dim() == 3 and x.shape[2] == 3
        B, N, _ = x.shape
        t = tfs.reshape(-1, 1, 4, 4)
        out = skinning(x, torch.ones(B, N, 1, device=x.device), t, inverse=inverse)
        outlier = None
        if verts is not None and inverse:
            outlier = _min_dist(x, verts) > self.max_dist
        return out, outlier

    __call__ = forward

    def forward_skinning(self, xc, cond, tfs):
        return self.forward(xc, tfs, inverse=False)[0]
