"""Articulation servers: MANO forward LBS and the rigid object transform, per FRAME (not per ray).

Mirrors the call surface of the reference's ``MANOServer`` / ``ObjectServer``
(code/src/model/mano/server.py:20-133, code/src/model/obj/server.py:19-56,
code/src/model/obj/object_model.py:12-70, code/src/utils/external/lbs.py:139-251).

These run on B <= ~50 frames of 778 vertices -- about 1e-5 of the path's FLOPs.  On the GPU
``MANOServer.forward`` is ONE launch of the fused HIP kernel ``hold_mano_lbs_fwd`` (csrc/mano.hip) with a
hand-derived backward ``hold_mano_lbs_bwd`` (pose / shape / translation gradients from d tfs and d verts).
The canonical-pose constants (verts_c, joints_c, tfs_c_inv; server.py:46-60) come from the same kernel, evaluated
lazily the first time they are needed on the module's device -- there is no torch / CPU implementation of the LBS in
this package (the CPU restatement lives in oracle/hold_oracle.py).
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn as nn

TIP_IDS = (744, 320, 443, 554, 671)


class VertexJointSelector(nn.Module):
    def __init__(self):
        super().__init__()
        self.register_buffer("extra_joints_idxs", torch.tensor(TIP_IDS, dtype=torch.long))


class ManoLayer(nn.Module):
    """Buffers named as the reference's MANO layer registers them (state_dict compatibility:
    ``...server.human_layer.{v_template,shapedirs,posedirs,J_regressor,parents,lbs_weights,hand_mean,pose_mean}``)."""

    def __init__(self, model: dict, is_rhand=True, dtype=torch.float32):
        super().__init__()
        def t(a):  # the licensed MANO pickle stores a scipy-sparse J_regressor and chumpy arrays
            a = a.toarray() if hasattr(a, "toarray") else a
            return torch.as_tensor(np.ascontiguousarray(np.asarray(a, dtype=np.float64)), dtype=dtype).contiguous()
        self.is_rhand = is_rhand
        self.faces = np.asarray(model["f"])
        self.register_buffer("faces_tensor", torch.as_tensor(self.faces.astype(np.int64)))
        self.register_buffer("v_template", t(model["v_template"]))
        self.register_buffer("shapedirs", t(np.asarray(model["shapedirs"])[:, :, :10]))
        self.register_buffer("J_regressor", t(model["J_regressor"]))
        pd = np.asarray(model["posedirs"])
        self.register_buffer("posedirs", t(np.reshape(pd, [-1, pd.shape[-1]]).T))
        parents = torch.as_tensor(np.asarray(model["kintree_table"])[0].astype(np.int64)).clone()
        parents[0] = -1
        self.register_buffer("parents", parents)
        self.bone_parents = np.asarray(model["kintree_table"])[0]
        self.register_buffer("lbs_weights", t(model["weights"]))
        hm = t(model["hands_mean"])
        self.register_buffer("hand_mean", hm)
        self.register_buffer("pose_mean", torch.cat([torch.zeros(3, dtype=dtype), hm]))
        self.vertex_joint_selector = VertexJointSelector()
        # member parameters the reference layer carries (unused by HOLD's forward, kept for checkpoints)
        for name, dim in [("betas", 10), ("global_orient", 3), ("body_pose", 3), ("transl", 3), ("hand_pose", 45)]:
            self.register_parameter(name, nn.Parameter(torch.zeros(1, dim, dtype=dtype), requires_grad=True))
        self._parent_list = [int(p) for p in parents.tolist()]


class _ManoLbsFn(torch.autograd.Function):
    """hold_mano_lbs_fwd / hold_mano_lbs_bwd (csrc/mano.hip): one launch per call, one workgroup per frame."""

    @staticmethod
    def forward(ctx, server, absolute, scene_scale, transl, thetas, betas):
        import ctypes as C

        from . import _lib

        hl = server.human_layer
        B = thetas.shape[0]
        dev = thetas.device
        mm = _lib.ManoModel()
        par = server._parents_i32
        for name, t in [("v_template", hl.v_template), ("shapedirs", hl.shapedirs), ("posedirs", hl.posedirs),
                        ("J_regressor", hl.J_regressor), ("parents", par), ("lbs_weights", hl.lbs_weights),
                        ("pose_mean", hl.pose_mean)]:
            assert t.is_cuda and t.is_contiguous()
            setattr(mm, name, t.data_ptr())
        mm.tfs_c_inv = None if absolute else server.tfs_c_inv.data_ptr()
        args = [a.detach().contiguous().float() for a in (betas, thetas, scene_scale.reshape(-1), transl)]
        verts = torch.empty(B, 778, 3, device=dev)
        jnts = torch.empty(B, 21, 3, device=dev)
        tfs = torch.empty(B, 16, 4, 4, device=dev)
        v_posed = torch.empty(B, 778, 3, device=dev)
        _lib.call("hold_mano_lbs_fwd", C.byref(mm), B, *[_lib.ptr(a) for a in args], _lib.ptr(verts), _lib.ptr(jnts),
                  _lib.ptr(tfs), _lib.ptr(v_posed))
        ctx.mm, ctx.args, ctx.keep = mm, args, (hl, server)
        ctx.mark_non_differentiable(v_posed)
        ctx.set_materialize_grads(False)  # unused outputs arrive as None in backward: no zero fill, no device read to test
        return verts, jnts, tfs, v_posed

    @staticmethod
    def backward(ctx, d_verts, d_jnts, d_tfs, _):
        import ctypes as C

        from . import _lib

        betas, thetas, scale, transl = ctx.args
        B = thetas.shape[0]
        dev = thetas.device
        if d_jnts is not None and bool((d_jnts != 0).any()):
            raise NotImplementedError("gradients through MANO joints are not used on the hot path")
        d_pose = torch.empty(B, 48, device=dev)
        d_betas = torch.empty(B, 10, device=dev)
        d_transl = torch.empty(B, 3, device=dev)
        _lib.call("hold_mano_lbs_bwd", C.byref(ctx.mm), B, *[_lib.ptr(a) for a in (betas, thetas, scale, transl)],
                  _lib.ptr(None if d_tfs is None else d_tfs.contiguous()),
                  _lib.ptr(None if d_verts is None else d_verts.contiguous()), _lib.ptr(d_pose), _lib.ptr(d_betas),
                  _lib.ptr(d_transl))
        return None, None, None, d_transl, d_pose, d_betas


class MANOServer(nn.Module):
    """GenericServer/MANOServer of code/src/model/mano/server.py:20-133.  ``model`` is the un-pickled MANO dict; when
    omitted it is read from ./body_models/MANO_{RIGHT,LEFT}.pkl as the reference does (server.py:121-128)."""

    def __init__(self, betas, is_rhand, model: dict = None):
        super().__init__()
        if model is None:
            import pickle
            with open(f"./body_models/MANO_{'RIGHT' if is_rhand else 'LEFT'}.pkl", "rb") as f:
                model = pickle.load(f, encoding="latin1")
        self.human_layer = ManoLayer(model, is_rhand)
        self.faces = self.human_layer.faces
        self.bone_parents = self.human_layer.bone_parents.astype(int)
        self.bone_parents[0] = -1
        self.bone_ids = [[int(self.bone_parents[i]), i] for i in range(16)]
        self.betas = None if betas is None else torch.as_tensor(np.asarray(betas), dtype=torch.float32)
        pc = torch.zeros(1, 62)
        pc[0, 0] = 1
        pc[0, 7:52] = -self.human_layer.hand_mean
        if self.betas is not None:
            pc[0, -10:] = self.betas
        self.param_canonical = pc
        self.cano_params = torch.split(pc, [1, 3, 48, 10], dim=1)
        self.register_buffer("_parents_i32", self.human_layer.parents.to(torch.int32), persistent=False)
        self._cano = None

    # ---- canonical pose constants (verts_c [1,778,3], joints_c [1,21,3], tfs_c_inv [16,4,4]) ----
    def _canonical(self):
        dev = self.human_layer.v_template.device
        if self._cano is None or self._cano["verts_c"].device != dev:
            if dev.type != "cuda":
                raise RuntimeError("hold_amd.MANOServer: the MANO LBS runs in libholdhip.so only -- move the module "
                                   "to the MI355X (.to('cuda')) before using its canonical vertices / transforms")
            with torch.no_grad():
                sc, tr, th, be = (a.to(dev) for a in self.cano_params)
                verts, jnts, tfs, _ = _ManoLbsFn.apply(self, True, sc, tr, th, be)
            self._cano = dict(verts_c=verts, joints_c=jnts, tfs_c_inv=tfs.squeeze(0).inverse().contiguous())
        return self._cano

    verts_c = property(lambda self: self._canonical()["verts_c"])
    joints_c = property(lambda self: self._canonical()["joints_c"])
    tfs_c_inv = property(lambda self: self._canonical()["tfs_c_inv"])

    def forward(self, scene_scale, transl, thetas, betas, absolute=False):
        hl = self.human_layer
        dev = hl.v_template.device
        scene_scale, transl, thetas, betas = (a.to(dev) for a in (scene_scale, transl, thetas, betas))
        verts, jnts, tfs, v_posed = _ManoLbsFn.apply(self, bool(absolute), scene_scale, transl, thetas, betas)
        return {"verts": verts, "jnts": jnts, "tfs": tfs, "v_posed": v_posed,
                "skin_weights": hl.lbs_weights[None].expand(verts.shape[0], -1, -1)}

    def forward_param(self, param_dict):
        get = lambda k: next(v for kk, v in param_dict.items() if k in kk)
        full_pose = torch.cat((get("global_orient"), get("pose")), dim=1)
        B = full_pose.shape[0]
        return self.forward(get("scene_scale").view(-1).repeat(B), get("transl"), full_pose, get("betas").repeat(B, 1))


_Q2M = {}


def _q2m_consts(device):
    """constants of the quaternion -> matrix map as two gathers of the outer product q q^T (flattened [16], order r i j k):
    entry e of the matrix = [e on the diagonal] + two_s * (sa[e] * qq[ia[e]] + sb[e] * qq[ib[e]])"""
    key = str(device)
    if key not in _Q2M:
        r, i, j, k = 0, 1, 2, 3
        P = lambda a, b: 4 * a + b
        terms = [((j, j, -1.0), (k, k, -1.0)), ((i, j, 1.0), (k, r, -1.0)), ((i, k, 1.0), (j, r, 1.0)),
                 ((i, j, 1.0), (k, r, 1.0)), ((i, i, -1.0), (k, k, -1.0)), ((j, k, 1.0), (i, r, -1.0)),
                 ((i, k, 1.0), (j, r, -1.0)), ((j, k, 1.0), (i, r, 1.0)), ((i, i, -1.0), (j, j, -1.0))]
        t = lambda v, dt: torch.tensor(v, dtype=dt, device=device)
        _Q2M[key] = (t([P(a[0], a[1]) for a, _ in terms], torch.long), t([a[2] for a, _ in terms], torch.float32),
                     t([P(b[0], b[1]) for _, b in terms], torch.long), t([b[2] for _, b in terms], torch.float32),
                     t([1.0, 0, 0, 0, 1.0, 0, 0, 0, 1.0], torch.float32))
    return _Q2M[key]


def axis_angle_to_matrix(aa):
    """common/rot.py:105-138,777-805 (axis-angle -> quaternion -> matrix).  The nine matrix entries are formed from two
    gathers of q q^T instead of entry by entry: the same products and sums (bit-identical results), a dozen launches
    instead of ~55 per call and direction."""
    angles = torch.norm(aa, p=2, dim=-1, keepdim=True)
    half = angles * 0.5
    small = angles.abs() < 1e-6
    soa = torch.where(small, 0.5 - (angles * angles) / 48,
                      torch.sin(half) / torch.where(small, torch.ones_like(angles), angles))
    q = torch.cat([torch.cos(half), aa * soa], -1)
    two_s = 2.0 / (q * q).sum(-1, keepdim=True)
    ia, sa, ib, sb, eye = _q2m_consts(aa.device)
    qq = (q.unsqueeze(-1) * q.unsqueeze(-2)).reshape(q.shape[:-1] + (16,))
    o = eye + two_s * (qq.index_select(-1, ia) * sa + qq.index_select(-1, ib) * sb)
    return o.reshape(q.shape[:-1] + (3, 3))


def _object_entity(entity_or_case):
    """the ``entities["object"]`` record of data.npy, given either the record itself or the sequence name the
    reference's ObjectModel / ObjectServer take (object_model.py:15-19)."""
    if isinstance(entity_or_case, str):
        return np.load(f"./data/{entity_or_case}/build/data.npy", allow_pickle=True).item()["entities"]["object"]
    return entity_or_case


class ObjectModel(nn.Module):
    def __init__(self, entity, template=None):
        super().__init__()
        entity = _object_entity(entity)
        if template is not None:
            entity = dict(entity, **{"pts.cano": np.asarray(template.vertices)})
        self.register_buffer("obj_scale", torch.tensor(np.array([entity["obj_scale"]]), dtype=torch.float32))
        self.register_buffer("v3d_cano", torch.as_tensor(entity["pts.cano"], dtype=torch.float32))
        nm = torch.as_tensor(entity["norm_mat"], dtype=torch.float32)
        self.register_buffer("norm_mat", nm)
        self.register_buffer("denorm_mat", torch.inverse(nm))

    def forward(self, rot, trans, scene_scale=None, want_verts=True):
        dev, dt = self.v3d_cano.device, self.v3d_cano.dtype
        B = rot.shape[0]
        scene_scale = torch.ones(B, device=dev) if scene_scale is None else scene_scale.view(B).to(dev)
        R = axis_angle_to_matrix(rot.to(dev)).view(B, 3, 3)
        top = torch.cat([R, trans.to(dev).view(B, 3, 1)], 2)
        if getattr(self, "_bottom", None) is None or self._bottom.device != dev or self._bottom.dtype != dt:
            self._bottom = torch.tensor([0.0, 0.0, 0.0, 1.0], device=dev, dtype=dt).view(1, 1, 4)  # one upload, not one per call
        bottom = self._bottom.expand(B, 1, 4)
        tf = torch.cat([top, bottom], 1)
        sm = torch.diag_embed(torch.cat([scene_scale[:, None].expand(B, 3), torch.ones(B, 1, device=dev)], 1))
        om = torch.diag(torch.cat([self.obj_scale.expand(3), torch.ones(1, device=dev)]))[None]
        tf = sm @ tf @ om @ self.denorm_mat[None]
        out = {"T": tf}
        if want_verts:
            vp = torch.cat([self.v3d_cano, torch.ones(self.v3d_cano.shape[0], 1, device=dev)], 1)
            v = torch.einsum("bij,nj->bni", tf, vp)
            out["vertices"] = v[:, :, :3] / v[:, :, 3:4]
        return out


class ObjectServer(nn.Module):
    """code/src/model/obj/server.py:19-56; ``entity`` = sequence name (reference signature) or the entity record."""

    def __init__(self, entity, template=None):
        super().__init__()
        self.object_model = ObjectModel(entity, template)

    verts_c = property(lambda self: self.object_model.v3d_cano[None])

    def forward(self, scene_scale, transl, thetas, absolute=False):
        o = self.object_model(rot=thetas, trans=transl, scene_scale=scene_scale)
        return {"verts": o["vertices"], "obj_tfs": o["T"][:, None, :, :]}

    def forward_param(self, param_dict):
        get = lambda k: next(v for kk, v in param_dict.items() if k in kk)
        go = get("global_orient")
        return self.forward(get("scene_scale").view(-1).repeat(go.shape[0]), get("transl"), go)
