"""Synthetic inputs for the hot path (no datasets, checkpoints or MANO files are available).

Everything here is plain numpy driven by ``np.random.RandomState(seed)`` so that the
oracle (CPU), the parity tests and ``bench.py`` (GPU box) regenerate bit-identical
inputs without shipping blobs.  Schemas follow what the reference reads:

* MANO pickle fields  -> code/src/utils/external/body_models.py:204,265-295,516-521,549-561
* ``data.npy`` entities -> code/src/model/mano/params.py:14-42, code/src/model/obj/params.py:9-29,
  code/src/model/obj/object_model.py:15-27
* cameras / rays      -> code/src/datasets/utils.py:230-282 (get_camera_params / lift)
"""
from __future__ import annotations

import numpy as np

MANO_PARENTS = np.array([-1, 0, 1, 2, 0, 4, 5, 0, 7, 8, 0, 10, 11, 0, 13, 14], dtype=np.int64)
NUM_VERTS = 778
NUM_JOINTS = 16
NUM_FACES = 1538
# fingertip vertex ids the reference appends as extra joints
# (code/src/utils/external/vertex_ids.py:69-75, order thumb,index,middle,ring,pinky)
MANO_TIP_IDS = np.array([744, 320, 443, 554, 671], dtype=np.int64)
# wrist-loop vertex ids the reference seals with a fan (common/body_models.py:46-49)
CIRCLE_V_ID = [108, 79, 78, 121, 214, 215, 279, 239, 234, 92, 38, 122, 118, 117, 119, 120]


def make_mano_model(is_rhand: bool = True, seed: int = 7) -> dict:
    """A deterministic MANO-shaped hand model (NOT the licensed MANO data).

    Returns the dict the reference un-pickles from ``./body_models/MANO_{RIGHT,LEFT}.pkl``.
    """
    rs = np.random.RandomState(seed + (0 if is_rhand else 1000))
    # --- rest joints: wrist + 5 fingers x 3 joints, metres, roughly MANO sized ---------
    joints = np.zeros((NUM_JOINTS, 3), dtype=np.float64)
    finger_dirs = np.array(
        [
            [0.95, 0.05, 0.30],  # index   (joints 1-3)
            [1.00, 0.02, 0.05],  # middle  (4-6)
            [0.85, 0.00, -0.40],  # pinky  (7-9)
            [0.97, 0.01, -0.18],  # ring   (10-12)
            [0.45, -0.10, 0.85],  # thumb  (13-15)
        ]
    )
    finger_dirs /= np.linalg.norm(finger_dirs, axis=1, keepdims=True)
    base_len = np.array([0.090, 0.092, 0.075, 0.086, 0.035])
    seg_len = np.array([[0.032, 0.022], [0.034, 0.024], [0.024, 0.017], [0.031, 0.022], [0.034, 0.027]])
    for f in range(5):
        j0 = 1 + 3 * f
        joints[j0] = finger_dirs[f] * base_len[f]
        joints[j0 + 1] = joints[j0] + finger_dirs[f] * seg_len[f, 0]
        joints[j0 + 2] = joints[j0 + 1] + finger_dirs[f] * seg_len[f, 1]
    if not is_rhand:
        joints[:, 2] *= -1.0

    # --- vertices: a closed, star-shaped "mitten" around the palm centre, open at the wrist ------------
    # Topology as MANO's: 778 vertices / 1538 faces = a triangulated sphere minus the fan of one valence-16 vertex,
    # whose ring is the wrist loop CIRCLE_V_ID that seal_mano_mesh closes again (common/body_models.py:46-73).
    # Directions: a Fibonacci lattice outside a cap around the wrist pole (-x) plus 16 ring directions on the cap's
    # inner circle; connectivity = convex hull of the directions (+ pole) with the pole's fan removed; positions =
    # centre + r(direction) * direction, so the surface is an embedded radial graph (no self-intersections).
    from scipy.spatial import ConvexHull
    ring_ids = np.array(CIRCLE_V_ID, dtype=np.int64)
    free_ids = np.array([i for i in range(NUM_VERTS) if i not in set(CIRCLE_V_ID)], dtype=np.int64)
    alpha = 0.25  # angular radius of the wrist ring around the pole
    m = len(free_ids)
    t = -np.cos(1.8 * alpha) + (1.0 + np.cos(1.8 * alpha)) * (np.arange(m) + 0.5) / m  # cos(angle to +x), cap excluded
    phi = np.arange(m) * np.pi * (3.0 - np.sqrt(5.0))
    st = np.sqrt(1.0 - t * t)
    dirs = np.zeros((NUM_VERTS + 1, 3))
    dirs[free_ids] = np.stack([t, st * np.cos(phi), st * np.sin(phi)], 1)
    # break the lattice's symmetry: on a regular lattice many query points are (nearly) equidistant from their 15th and
    # 16th nearest vertex, and which one a K = 15 neighbour search keeps then depends on the last bit of the distance
    # arithmetic (fma contraction, summation order) -- real MANO vertices are irregular
    jit = np.random.RandomState(4242 + (0 if is_rhand else 1)).normal(scale=0.02, size=(m, 3))
    dirs[free_ids] += jit
    dirs[free_ids] /= np.linalg.norm(dirs[free_ids], axis=1, keepdims=True)
    az = 2.0 * np.pi * np.arange(16) / 16.0
    dirs[ring_ids] = np.stack([-np.cos(alpha) * np.ones(16), np.sin(alpha) * np.cos(az), np.sin(alpha) * np.sin(az)], 1)
    dirs[NUM_VERTS] = [-1.0, 0.0, 0.0]
    hull = ConvexHull(dirs)
    faces = hull.simplices.astype(np.int64)
    assert faces.shape[0] == 2 * (NUM_VERTS + 1) - 4, faces.shape
    tri = dirs[faces]
    flip = np.einsum("fi,fi->f", np.cross(tri[:, 1] - tri[:, 0], tri[:, 2] - tri[:, 0]), tri.mean(1)) < 0
    faces[flip] = faces[flip][:, [0, 2, 1]]  # outward orientation
    fan = (faces == NUM_VERTS).any(1)
    assert fan.sum() == 16 and set(faces[fan].ravel()) - {NUM_VERTS} == set(CIRCLE_V_ID)
    # the reference's seal faces are [ring[i-1], ring[i], 778]: the ring must run so that they face outward too
    f0 = [f for f in faces[fan] if ring_ids[0] in f and ring_ids[1] in f][0].tolist()
    k = f0.index(int(ring_ids[0]))
    if f0[(k + 1) % 3] != int(ring_ids[1]):  # hull fan runs the other way round: mirror the ring's azimuth
        dirs[ring_ids, 2] *= -1.0
        dirs[free_ids, 2] *= -1.0
        faces = faces[:, [0, 2, 1]]
    faces = faces[~fan]
    assert faces.shape[0] == NUM_FACES
    ax = np.array([0.085, 0.022, 0.070])
    d = dirs[:NUM_VERTS]
    r = 1.0 / np.sqrt(((d / ax) ** 2).sum(1))
    r *= 1.0 + 0.10 * np.sin(3.0 * np.arctan2(d[:, 2], d[:, 0])) * (1.0 - d[:, 1] ** 2)
    verts = np.array([0.045, 0.0, 0.01]) + d * r[:, None]
    if not is_rhand:  # mirrored hand: z -> -z flips the orientation, so swap two corners of every face
        verts[:, 2] *= -1.0
        faces = faces[:, [1, 0, 2]]
    tips = [verts[vid] for vid in MANO_TIP_IDS]

    # --- skinning weights: soft assignment to nearest joints, rows sum to 1 ------------
    d2 = ((verts[:, None, :] - joints[None, :, :]) ** 2).sum(-1)
    w = np.exp(-d2 / (2 * 0.018**2))
    keep = np.argsort(-w, axis=1)[:, :4]
    mask = np.zeros_like(w)
    np.put_along_axis(mask, keep, 1.0, axis=1)
    w = w * mask + 1e-12 * mask
    w /= w.sum(1, keepdims=True)

    # --- joint regressor: normalised gaussian around each joint ------------------------
    jr = np.exp(-d2.T / (2 * 0.012**2)) + 1e-9
    jr /= jr.sum(1, keepdims=True)

    # --- blend shapes ------------------------------------------------------------------
    shapedirs = rs.normal(scale=0.0025, size=(NUM_VERTS, 3, 10))
    posedirs = rs.normal(scale=0.0004, size=(NUM_VERTS, 3, 135))
    hands_components = np.linalg.qr(rs.normal(size=(45, 45)))[0]
    hands_mean = rs.normal(scale=0.12, size=45)

    kintree = np.stack([MANO_PARENTS.copy(), np.arange(NUM_JOINTS)], axis=0).astype(np.int64)
    kintree[0, 0] = 4294967295  # as in the MANO pickle; reference overwrites with -1
    return {
        "v_template": verts.astype(np.float64),
        "shapedirs": shapedirs,
        "posedirs": posedirs,
        "J_regressor": jr,
        "kintree_table": kintree,
        "weights": w,
        "f": faces,
        "hands_components": hands_components,
        "hands_mean": hands_mean,
    }


def look_at_c2w(cam_pos: np.ndarray, target=np.zeros(3)) -> np.ndarray:
    """camera-to-world 4x4 with +z forward, +y down (pin-hole convention of lift())."""
    fwd = target - cam_pos
    fwd = fwd / np.linalg.norm(fwd)
    up = np.array([0.0, -1.0, 0.0])
    right = np.cross(fwd, up)
    if np.linalg.norm(right) < 1e-6:
        right = np.array([1.0, 0.0, 0.0])
    right /= np.linalg.norm(right)
    down = np.cross(fwd, right)
    c2w = np.eye(4)
    c2w[:3, 0] = right
    c2w[:3, 1] = down
    c2w[:3, 2] = fwd
    c2w[:3, 3] = cam_pos
    return c2w


def make_scene(n_frames: int = 4, two_hands: bool = False, seed: int = 1,
               scene_bounding_sphere: float = 6.0) -> dict:
    """Per-frame poses / cameras of the synthetic sequence."""
    rs = np.random.RandomState(seed)
    sc = {"n_frames": n_frames, "scene_bounding_sphere": float(scene_bounding_sphere),
          "scene_scale": 3.0, "entities": {}}
    hands = ["right", "left"] if two_hands else ["right"]
    for hi, h in enumerate(hands):
        poses = rs.normal(scale=0.2, size=(n_frames, 48))
        trans = rs.normal(scale=0.02, size=(n_frames, 3)) + np.array([0.05 - 0.12 * hi, 0.0, 0.05])
        sc["entities"][h] = {
            "hand_poses": poses.astype(np.float32),
            "hand_trans": trans.astype(np.float32),
            "mean_shape": rs.normal(scale=0.5, size=10).astype(np.float32),
        }
    obj_rot = rs.normal(scale=0.3, size=(n_frames, 3))
    obj_t = rs.normal(scale=0.03, size=(n_frames, 3)) + np.array([-0.05, 0.02, -0.02])
    pts = rs.normal(size=(2000, 3))
    # radius 0.45: the object's canonical point cloud bounds the canonical-meshing box (object_node.py:49-50: bbox x 2,
    # then x 1.1), which has to contain the zero level set of the geometric-init SDF (radius up to ~0.8)
    pts = pts / np.linalg.norm(pts, axis=1, keepdims=True) * 0.45
    sc["entities"]["object"] = {
        "object_poses": np.concatenate([obj_rot, obj_t], 1).astype(np.float32),
        "pts.cano": pts.astype(np.float32),
        "obj_scale": np.float32(1.0),
        "norm_mat": np.eye(4, dtype=np.float32),
    }
    cams = []
    for f in range(n_frames):
        ang = 0.35 * f
        cam_pos = 2.0 * np.array([np.sin(ang) * 0.6, -0.2 + 0.1 * f / max(n_frames - 1, 1), -np.cos(ang * 0.6)])
        cam_pos = cam_pos / np.linalg.norm(cam_pos) * 2.0
        cams.append(look_at_c2w(cam_pos))
    sc["c2w"] = np.stack(cams).astype(np.float32)
    return sc


def make_intrinsics(width: int, height: int) -> np.ndarray:
    K = np.eye(4, dtype=np.float32)
    K[0, 0] = K[1, 1] = 1.2 * width
    K[0, 2] = width / 2.0
    K[1, 2] = height / 2.0
    return K


def make_uv(width: int, height: int) -> np.ndarray:
    """pixel grid exactly as code/src/datasets/image_dataset.py:66-67 builds it (x fast)."""
    ys, xs = np.meshgrid(np.arange(height), np.arange(width), indexing="ij")
    uv = np.stack([xs, ys], axis=-1).reshape(-1, 2).astype(np.float32)
    return uv


def make_batch(scene: dict, frame_ids, uv: np.ndarray, width: int, height: int, seed: int = 1) -> dict:
    """The per-call input dict HOLDNet.forward consumes (code/src/hold/hold.py:110-124,169-179),
    as numpy; callers convert to torch tensors on their device.  uv: [n_pix,2] shared by frames."""
    frame_ids = np.asarray(frame_ids, dtype=np.int64)
    B = len(frame_ids)
    rs = np.random.RandomState(seed + 99)
    K = make_intrinsics(width, height)
    batch = {
        "uv": np.broadcast_to(uv[None], (B,) + uv.shape).copy(),
        "intrinsics": np.broadcast_to(K[None], (B, 4, 4)).copy(),
        "extrinsics": scene["c2w"][frame_ids].copy(),
        "idx": frame_ids,
        "gt.rgb": rs.uniform(size=(B, uv.shape[0], 3)).astype(np.float32),
        "gt.mask": rs.choice(np.array([0, 50, 150, 250]), size=(B, uv.shape[0])).astype(np.int64),
    }
    for name in scene["entities"]:
        p = np.zeros((B, 62 if name != "object" else 7), dtype=np.float32)
        p[:, 0] = scene["scene_scale"]
        batch[f"{name}.params"] = p
    return batch


# ------------------------------------------------------------------------------------------
# Network weights (reference state_dict names; shapes from code/confs/general.yaml and the
# constructors code/src/networks/shape_net.py:9-82, texture_net.py:8-44)
# ------------------------------------------------------------------------------------------
FG_MULTIRES = 6
BG_MULTIRES = 10
BG_VIEW_MULTIRES = 4
FEAT = 256
TIME_CODE = 32


def _implicit_dims(d_in, multires, cond_dim):
    e = d_in + d_in * 2 * multires
    dims = [e] + [256] * 8 + [1 + FEAT]
    layers = []
    for l in range(9):
        out = dims[l + 1] - dims[0] if (l + 1) == 4 else dims[l + 1]
        inn = dims[l] + (cond_dim if l == 0 else 0)
        layers.append((out, inn))
    return e, layers


def _linear_default(rs, out, inn):
    bound = 1.0 / np.sqrt(max(inn, 1))
    return (rs.uniform(-bound, bound, size=(out, inn)).astype(np.float32),
            rs.uniform(-bound, bound, size=(out,)).astype(np.float32))


def make_state_dict(scene: dict, seed: int = 1, perturb: float = 0.02, barf_iter: int = 3999) -> dict:
    """name -> np.ndarray for every learnable tensor of HOLDNet (reference naming, SURVEY.md 5
    'Checkpoint' row).  fg SDF nets follow the geometric init of shape_net.py:51-72 (a sphere-like SDF whose zero level
    set has radius ~0.25, exactly what the reference constructor produces) plus a perturbation of the feature rows."""
    rs = np.random.RandomState(seed + 31337)
    sd = {}
    F = scene["n_frames"]
    for name, ent in scene["entities"].items():
        is_obj = name == "object"
        cond = 0 if is_obj else 45
        e, layers = _implicit_dims(3, FG_MULTIRES, cond)
        pfx = f"nodes.{name}.implicit_network."
        for l, (out, inn) in enumerate(layers):
            std = np.sqrt(2) / np.sqrt(out)
            if l == 8:
                w = rs.normal(np.sqrt(np.pi) / np.sqrt(inn), 1e-4, size=(out, inn))
                # -0.35 (the config's init uses 0.6): level sets of radius ~0.15 (hand) / ~0.25 (object) in canonical space,
                # i.e. blobs of hand / object size in front of the camera instead of spheres that swallow it
                b = np.full((out,), -0.35)
            elif l == 0:
                w = np.zeros((out, inn))
                w[:, :3] = rs.normal(0.0, std, size=(out, 3))
                b = np.zeros((out,))
            elif l == 4:
                w = rs.normal(0.0, std, size=(out, inn))
                w[:, -(e - 3):] = 0.0
                b = np.zeros((out,))
            else:
                w = rs.normal(0.0, std, size=(out, inn))
                b = np.zeros((out,))
            g = np.linalg.norm(w, axis=1, keepdims=True)
            # the trunk and the sdf row stay at the geometric init, so the zero level set (a sphere of radius ~0.25 in
            # canonical space, as the reference's own constructor gives) survives; only the 256 FEATURE rows of the
            # last layer are perturbed so that features / colours are not all alike.  (Perturbing every weight_v, as
            # SURVEY 8(d) first planned, lifts the whole SDF above zero: softplus units are rectifiers, zero-mean
            # weight noise has a positive mean effect -- the scene became fog without a surface.)
            v = w.copy()
            if l == 8:
                v[1:] += rs.normal(0.0, 2.5 * perturb, size=v[1:].shape)
            sd[pfx + f"lin{l}.weight_g"] = g.astype(np.float32)
            sd[pfx + f"lin{l}.weight_v"] = v.astype(np.float32)
            sd[pfx + f"lin{l}.bias"] = (b + rs.normal(0, 0.005, size=b.shape) * (l < 8)).astype(np.float32)
        pfx = f"nodes.{name}.rendering_network."
        w, b = _linear_default(rs, 8, cond)
        sd[pfx + "lin_pose.weight"], sd[pfx + "lin_pose.bias"] = w, b
        d0 = 3 + 3 + 8 + FEAT + (TIME_CODE if is_obj else 0)
        rdims = [d0, 256, 256, 256, 256, 3]
        for l in range(5):
            w, b = _linear_default(rs, rdims[l + 1], rdims[l])
            sd[pfx + f"lin{l}.weight_g"] = np.linalg.norm(w, axis=1, keepdims=True).astype(np.float32)
            sd[pfx + f"lin{l}.weight_v"] = (w * rs.uniform(0.8, 1.6)).astype(np.float32)
            sd[pfx + f"lin{l}.bias"] = b
        sd[f"nodes.{name}.density.beta"] = np.float32(0.1)
        if is_obj:
            sd["nodes.object.params.global_orient.weight"] = ent["object_poses"][:, :3].copy()
            sd["nodes.object.params.transl.weight"] = ent["object_poses"][:, 3:].copy()
            sd["nodes.object.frame_latent_encoder.weight"] = rs.normal(0, 0.5, size=(F, TIME_CODE)).astype(np.float32)
            sd["nodes.object.implicit_network.embedder_obj.alpha_iter"] = np.int64(barf_iter)
        else:
            sd[f"nodes.{name}.params.betas.weight"] = ent["mean_shape"][None].copy()
            sd[f"nodes.{name}.params.global_orient.weight"] = ent["hand_poses"][:, :3].copy()
            sd[f"nodes.{name}.params.pose.weight"] = ent["hand_poses"][:, 3:].copy()
            sd[f"nodes.{name}.params.transl.weight"] = ent["hand_trans"].copy()
    # background (no weight norm, default nn.Linear init; shape_net.py with d_in=4, L=10, cond=frame 32)
    e, layers = _implicit_dims(4, BG_MULTIRES, TIME_CODE)
    for l, (out, inn) in enumerate(layers):
        w, b = _linear_default(rs, out, inn)
        sd[f"background.bg_implicit_network.lin{l}.weight"] = (w * 1.5).astype(np.float32)
        sd[f"background.bg_implicit_network.lin{l}.bias"] = b
    dv = 3 + 3 * 2 * BG_VIEW_MULTIRES
    w, b = _linear_default(rs, 128, dv + TIME_CODE + FEAT)
    sd["background.bg_rendering_network.lin0.weight"], sd["background.bg_rendering_network.lin0.bias"] = w, b
    w, b = _linear_default(rs, 3, 128)
    sd["background.bg_rendering_network.lin1.weight"], sd["background.bg_rendering_network.lin1.bias"] = w, b
    sd["background.frame_latent_encoder.weight"] = rs.normal(0, 0.5, size=(F, TIME_CODE)).astype(np.float32)
    return sd
