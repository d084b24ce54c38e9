"""TEST INFRASTRUCTURE ONLY -- CPU restatement (plain torch ops, fp32 or fp64) of the
reference's volumetric hand-object rendering hot path (zc-alexfan/hold).

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg import
this module, and only as the checker.  The product path (``hold_amd``) never does.

Parity status: PINNED -- ``tests/test_oracle_golden.py`` checks every function here against
fixtures produced by running the reference's own Python on CPU (``scripts/make_golden.py``,
fixtures in ``tests/golden/``).  The reference has no tests / golden vectors of its own
(SURVEY.md 4).  Third-party arithmetic restated from its published behaviour:
pytorch3d 0.7.4 ``ops.knn_points`` (squared L2, K smallest, ascending) -- parity unpinned for
that dependency itself, anchored on the reference call site code/src/model/mano/deformer.py:85-87.

Each function cites the reference lines it follows (paths relative to /root/reference).
All functions are written functionally over a flat ``sd`` (reference state_dict names).
"""
from __future__ import annotations

import math

import numpy as np
import torch
import torch.nn.functional as F

# ------------------------------------------------------------------------------------------
# networks
# ------------------------------------------------------------------------------------------


def effective_weight(sd, prefix):
    """weight-normed Linear (torch.nn.utils.weight_norm, dim=0): w = g * v / ||v||_row.
    code/src/networks/shape_net.py:79-80, texture_net.py:40-41."""
    if prefix + ".weight_g" in sd:
        v = sd[prefix + ".weight_v"]
        g = sd[prefix + ".weight_g"]
        return v * (g / v.norm(dim=1, keepdim=True))
    return sd[prefix + ".weight"]


def barf_weights(alpha_iter: int, num_freq: int, input_dims: int, start=1000, end=10000):
    """code/src/engine/embedders.py:72-105 (alphas table + compute_barf_weights)."""
    alphas = torch.cat((torch.zeros(start), torch.linspace(0, num_freq, end - start)), 0)
    alpha = alphas[alpha_iter]
    k = torch.arange(num_freq, dtype=torch.float32)
    ak = alpha - k
    w = torch.clamp(ak, 0, 1)
    cos_idx = torch.logical_and(0 <= ak, ak < 1)
    cos_val = (1 - torch.cos(ak * math.pi)) / 2
    w[cos_idx] = cos_val[cos_idx]
    w = w[:, None].repeat(1, input_dims * 2).view(-1)
    return torch.cat((torch.ones(input_dims), w), 0)


def embed(x, multires: int, weights=None):
    """code/src/engine/embedders.py:18-50: [x, sin(2^0 x), cos(2^0 x), sin(2^1 x), ...]."""
    out = [x]
    for k in range(multires):
        f = 2.0 ** k
        out.append(torch.sin(x * f))
        out.append(torch.cos(x * f))
    e = torch.cat(out, -1)
    if weights is not None:
        e = e * weights.to(e.dtype)[None, :]
    return e


def softplus100(x):
    return F.softplus(x, beta=100)


def implicit_net(sd, prefix, x, cond, multires, embed_w=None, zero_cond=False):
    """code/src/networks/shape_net.py:84-130.  x [P,d_in]; cond [P,C] or None.
    Returns [P, 1+256].  zero_cond: the MANO pose condition is multiplied by 0 (:104-106)."""
    e = embed(x, multires, embed_w)
    h = e
    for l in range(9):
        W = effective_weight(sd, f"{prefix}.lin{l}")
        b = sd[f"{prefix}.lin{l}.bias"]
        if l == 0 and cond is not None:
            h = torch.cat([h, cond * 0.0 if zero_cond else cond], -1)
        if l == 4:
            h = torch.cat([h, e], 1) / np.sqrt(2)
        h = F.linear(h, W, b)
        if l < 8:
            h = softplus100(h)
    return h


def rendering_net_pose(sd, prefix, points, normals, body_pose, feats):
    """code/src/networks/texture_net.py:46-101, mode 'pose'.  body_pose [P,45] or [P,0]."""
    if body_pose.shape[1] > 0:
        bp = F.linear(body_pose, sd[f"{prefix}.lin_pose.weight"], sd[f"{prefix}.lin_pose.bias"])
    else:
        bp = torch.zeros(points.shape[0], 8, dtype=points.dtype)
    x = torch.cat([points, normals, bp, feats], -1)
    for l in range(5):
        x = F.linear(x, effective_weight(sd, f"{prefix}.lin{l}"), sd[f"{prefix}.lin{l}.bias"])
        if l < 4:
            x = torch.relu(x)
    return torch.sigmoid(x)


def rendering_net_bg(sd, prefix, view_dirs, frame_latent, feats):
    """texture_net.py:55-70,94-101, mode 'nerf_frame_encoding' (view embed L=4)."""
    x = torch.cat([embed(view_dirs, 4), frame_latent, feats], -1)
    x = torch.relu(F.linear(x, sd[f"{prefix}.lin0.weight"], sd[f"{prefix}.lin0.bias"]))
    x = F.linear(x, sd[f"{prefix}.lin1.weight"], sd[f"{prefix}.lin1.bias"])
    return torch.sigmoid(x)


# ------------------------------------------------------------------------------------------
# density / sampler
# ------------------------------------------------------------------------------------------


def laplace_density(sdf, beta):
    """code/src/engine/density.py:21-26."""
    alpha = 1 / beta
    return alpha * (0.5 + 0.5 * sdf.sign() * torch.expm1(-sdf.abs() / beta))


def get_beta(sd, node):
    """density.py:28-30 (beta_min 1e-4, general.yaml:65-68)."""
    return sd[f"nodes.{node}.density.beta"].abs() + 1e-4


def sphere_far(cam_loc, ray_dirs, r):
    """code/src/engine/ray_sampler.py:6-25 (second root, clamped at 0)."""
    rcd = (ray_dirs * cam_loc).sum(-1, keepdim=True)
    under = rcd ** 2 - (cam_loc.norm(2, 1, keepdim=True) ** 2 - r ** 2)
    assert (under > 0).all(), "BOUNDING SPHERE PROBLEM"
    return (torch.sqrt(under) - rcd).clamp_min(0.0)


def uniform_z(near, far, n, t_rand=None):
    """ray_sampler.py:54-80.  t_rand (training) is the torch.rand draw of :76."""
    t = torch.linspace(0.0, 1.0, steps=n, dtype=far.dtype)
    z = near * (1.0 - t) + far * t
    if t_rand is not None:
        mids = 0.5 * (z[..., 1:] + z[..., :-1])
        upper = torch.cat([mids, z[..., -1:]], -1)
        lower = torch.cat([z[..., :1], mids], -1)
        z = lower + (upper - lower) * t_rand
    return z


def error_bound(beta, sdf, dists, d_star):
    """ray_sampler.py:354-366.  beta scalar or [N,1]; sdf [N,S]; dists, d_star [N,S-1]."""
    density = laplace_density(sdf, beta)
    sfe = torch.cat([torch.zeros(dists.shape[0], 1, dtype=sdf.dtype), dists * density[:, :-1]], -1)
    integral = torch.cumsum(sfe, -1)
    eps_sec = torch.exp(-d_star / beta) * (dists ** 2.0) / (4 * beta ** 2)
    e_int = torch.cumsum(eps_sec, -1)
    bound = (torch.clamp(torch.exp(e_int), max=1.0e6) - 1.0) * torch.exp(-integral[:, :-1])
    return bound.max(-1)[0]


def d_star_bound(z_vals, sdf):
    """ray_sampler.py:191-206 (Theorem 1 triangle bound)."""
    d = sdf
    dists = z_vals[:, 1:] - z_vals[:, :-1]
    a, b, c = dists, d[:, :-1].abs(), d[:, 1:].abs()
    first = a.pow(2) + b.pow(2) <= c.pow(2)
    second = a.pow(2) + c.pow(2) <= b.pow(2)
    d_star = torch.zeros_like(dists)
    d_star[first] = b[first]
    d_star[second] = c[second]
    s = (a + b + c) / 2.0
    area = s * (s - a) * (s - b) * (s - c)
    mask = ~first & ~second & (b + c - a > 0)
    d_star[mask] = (2.0 * torch.sqrt(area[mask])) / (a[mask])
    d_star = (d[:, 1:].sign() * d[:, :-1].sign() == 1) * d_star
    return dists, d_star


def inv_cdf(cdf, bins, u):
    """ray_sampler.py:295-307."""
    inds = torch.searchsorted(cdf, u, right=True)
    below = torch.clamp(inds - 1, min=0)
    above = torch.clamp(inds, max=cdf.shape[-1] - 1)
    c0, c1 = torch.gather(cdf, 1, below), torch.gather(cdf, 1, above)
    b0, b1 = torch.gather(bins, 1, below), torch.gather(bins, 1, above)
    denom = c1 - c0
    denom = torch.where(denom < 1e-5, torch.ones_like(denom), denom)
    t = (u - c0) / denom
    return b0 + t * (b1 - b0)


def sampler_round(z_vals, sdf, beta, beta0, eps=0.1, beta_iters=10):
    """One pass of ray_sampler.py:191-244 given the merged sdf at z_vals.
    Returns (beta_new[N], dists, d_star, weights, transmittance)."""
    dists, d_star = d_star_bound(z_vals, sdf)
    beta = beta.clone()
    cur = error_bound(beta0, sdf, dists, d_star)
    beta[cur <= eps] = beta0
    bmin, bmax = beta0 * torch.ones_like(beta), beta
    for _ in range(beta_iters):
        bmid = (bmin + bmax) / 2.0
        cur = error_bound(bmid.unsqueeze(-1), sdf, dists, d_star)
        bmax = torch.where(cur <= eps, bmid, bmax)
        bmin = torch.where(cur > eps, bmid, bmin)
    beta = bmax
    density = laplace_density(sdf, beta.unsqueeze(-1))
    dists_e = torch.cat([dists, torch.full((dists.shape[0], 1), 1e10, dtype=sdf.dtype)], -1)
    fe = dists_e * density
    sfe = torch.cat([torch.zeros(dists.shape[0], 1, dtype=sdf.dtype), fe[:, :-1]], -1)
    alpha = 1 - torch.exp(-fe)
    trans = torch.exp(-torch.cumsum(sfe, -1))
    weights = alpha * trans
    return beta, dists, d_star, weights, trans


def beta_search_conditioning(z_vals, sdf, beta, beta0, eps=0.1, beta_iters=10, rel_noise=1e-4):
    """fp64 run of the beta line search of ray_sampler.py:207-220 (sampler_round above) that also reports which rays are
    BORDERLINE: the search takes 1 + beta_iters threshold decisions `error_bound(beta_mid) <= eps` per ray, and a ray whose error
    bound comes within rel_noise * eps of the threshold at one of them is decided by the rounding of whoever evaluates the bound
    (exp of a 640-term cumulative sum, minus one: fp32 evaluations of the same bound differ by ~1e-5 relative) -- a different
    decision moves beta by up to half the current bracket.  Every other ray follows the same bracket sequence in any fp32-class
    arithmetic, and its beta is defined to rounding.  Returns (beta64 [N], borderline [N] bool, margin [N]) with margin = the
    smallest |error bound - eps| / eps over the decisions the fp64 search took."""
    z, s = z_vals.double(), sdf.double()
    b = beta.double().reshape(-1).clone()
    beta0 = float(beta0)
    dists, d_star = d_star_bound(z, s)
    cur = error_bound(beta0, s, dists, d_star)
    margin = (cur - eps).abs() / eps
    b[cur <= eps] = beta0
    bmin, bmax = beta0 * torch.ones_like(b), b
    for _ in range(beta_iters):
        bmid = (bmin + bmax) / 2.0
        cur = error_bound(bmid.unsqueeze(-1), s, dists, d_star)
        live = bmax > bmin  # a ray already at beta0 re-decides at beta0 itself: the same decision as the first
        margin = torch.where(live, torch.minimum(margin, (cur - eps).abs() / eps), margin)
        bmax = torch.where(cur <= eps, bmid, bmax)
        bmin = torch.where(cur > eps, bmid, bmin)
    return bmax, margin <= rel_noise, margin


def round_cdf(z_vals, sdf, beta, more, add_tiny=1e-6):
    """the CDF a round samples from (ray_sampler.py:247-293), given the round's final beta: the error-bound pdf while the
    hierarchy grows (`more`), the rendering weights + 1e-5 in the last round.  Any float dtype (the parity tests evaluate
    it in fp64 to bound the conditioning of the inverse-CDF stage)."""
    dists, d_star = d_star_bound(z_vals, sdf)
    density = laplace_density(sdf, beta.unsqueeze(-1))
    dists_e = torch.cat([dists, torch.full((dists.shape[0], 1), 1e10, dtype=sdf.dtype)], -1)
    fe = dists_e * density
    sfe = torch.cat([torch.zeros(dists.shape[0], 1, dtype=sdf.dtype), fe[:, :-1]], -1)
    trans = torch.exp(-torch.cumsum(sfe, -1))
    if more:
        eps_sec = torch.exp(-d_star / beta.unsqueeze(-1)) * (dists ** 2.0) / (4 * beta.unsqueeze(-1) ** 2)
        bo = (torch.clamp(torch.exp(torch.cumsum(eps_sec, -1)), max=1.0e6) - 1.0) * trans[:, :-1]
        pdf = bo + add_tiny
    else:
        pdf = ((1 - torch.exp(-fe)) * trans)[..., :-1] + 1e-5
    pdf = pdf / pdf.sum(-1, keepdim=True)
    cdf = torch.cumsum(pdf, -1)
    return torch.cat([torch.zeros_like(cdf[..., :1]), cdf], -1)


def inv_cdf_conditioning(z_vals, sdf, beta, more, u, eps_c, add_tiny=1e-6):
    """fp64 inverse-CDF samples of a round and, per sample, how far it moves when its CDF argument moves by +- eps_c (the
    size of fp32 rounding differences between two evaluations of the same CDF): (z64[N,n], spread[N,n]).  Where the CDF
    is flat (no mass: empty space behind / in front of the surface) a sample is defined only up to the width of the flat
    stretch -- that is the discontinuity of the stage, and `spread` measures it sample by sample."""
    z, s, b, u = z_vals.double(), sdf.double(), beta.double(), u.double()
    cdf = round_cdf(z, s, b, more, add_tiny)
    z0 = inv_cdf(cdf, z, u.contiguous())
    lo = inv_cdf(cdf, z, torch.clamp(u - eps_c, min=0.0).contiguous())
    hi = inv_cdf(cdf, z, torch.clamp(u + eps_c, max=1.0).contiguous())
    # u = 1 exactly sits on the end of the CDF: an evaluation whose last knot rounds above 1 finds it inside the last
    # stretch of mass, one whose last knot is <= 1 returns the last bin edge
    last = z[:, -1:].expand_as(z0)
    hi = torch.where(u >= 1.0 - eps_c, torch.maximum(hi, last), hi)
    return z0, torch.maximum(hi, z0) - torch.minimum(lo, z0)


def error_bound_sample(z_vals, sdf_fn, cam_loc, ray_dirs, beta0, R, is_training=False, rng=None,
                       N_samples=64, N_eval=128, N_extra=32, eps=0.1, beta_iters=10, max_iters=5,
                       add_tiny=1e-6, near=0.0, trace=None, sync=None):
    """ErrorBoundSampler.get_z_vals, code/src/engine/ray_sampler.py:128-352.
    z_vals: the initial uniform samples; sdf_fn(points[P,3]) -> sdf[P].  ``rng`` supplies the
    training-mode draws (dict with 'u_final' [N,N_samples], 'perm' indices)."""
    N = z_vals.shape[0]
    dt = z_vals.dtype
    samples, samples_idx = z_vals, None
    dists = z_vals[:, 1:] - z_vals[:, :-1]
    bound = (1.0 / (4.0 * math.log(eps + 1.0))) * (dists ** 2.0).sum(-1)
    beta = torch.sqrt(bound)
    total_iters, not_converge = 0, True
    sdf = None
    while not_converge and total_iters < max_iters:
        pts = cam_loc.unsqueeze(1) + samples.unsqueeze(2) * ray_dirs.unsqueeze(1)
        with torch.no_grad():
            s_sdf = sdf_fn(pts.reshape(-1, 3)).reshape(N, -1)
        if samples_idx is not None:
            sdf = torch.gather(torch.cat([sdf, s_sdf], -1), 1, samples_idx)
        else:
            sdf = s_sdf
        beta, dists, d_star, weights, trans = sampler_round(z_vals, sdf, beta, beta0, eps, beta_iters)
        total_iters += 1
        bmax = float(beta.max())
        if sync is not None:  # data-parallel shards: MAX over the ranks (hold_amd.sampler.ErrorBoundSampler.sync_round)
            bmax = sync(bmax)
        not_converge = bool(bmax > float(beta0))
        more = not_converge and total_iters < max_iters
        n_new = N_eval if more else N_samples
        cdf = round_cdf(z_vals, sdf, beta, more, add_tiny)
        if more or not is_training:
            u = torch.linspace(0.0, 1.0, steps=n_new, dtype=dt).unsqueeze(0).repeat(N, 1)
        else:
            u = rng["u_final"].to(dt)
        samples = inv_cdf(cdf, z_vals, u.contiguous())
        if trace is not None:
            trace.append(dict(z_vals=z_vals.clone(), sdf=sdf.clone(), beta=beta.clone(),
                              samples=samples.clone(), more=more))
        if more:
            z_vals, samples_idx = torch.sort(torch.cat([z_vals, samples], -1), -1)
    z_samples = samples
    near_t = near * torch.ones(N, 1, dtype=dt)
    far = sphere_far(cam_loc, ray_dirs, R)
    if is_training:
        perm = rng["perm"]
        idx = (perm(z_vals.shape[1]) if callable(perm) else perm)[:N_extra]
    else:
        idx = torch.linspace(0, z_vals.shape[1] - 1, N_extra).long()
    extra = torch.cat([near_t, far, z_vals[:, idx]], -1)
    z_out, _ = torch.sort(torch.cat([z_samples, extra], -1), -1)
    return z_out, total_iters


# ------------------------------------------------------------------------------------------
# MANO forward LBS and the servers
# ------------------------------------------------------------------------------------------


def batch_rodrigues(rot_vecs):
    """code/src/utils/external/lbs.py:298-330 (note norm(rot_vecs + 1e-8) at :313)."""
    n = rot_vecs.shape[0]
    angle = torch.norm(rot_vecs + 1e-8, dim=1, keepdim=True)
    rot_dir = rot_vecs / angle
    cos = torch.cos(angle)[:, None]
    sin = torch.sin(angle)[:, None]
    rx, ry, rz = torch.split(rot_dir, 1, dim=1)
    z = torch.zeros((n, 1), dtype=rot_vecs.dtype)
    K = torch.cat([z, -rz, ry, rz, z, -rx, -ry, rx, z], 1).view(n, 3, 3)
    ident = torch.eye(3, dtype=rot_vecs.dtype)[None]
    return ident + sin * K + (1 - cos) * torch.bmm(K, K)


def mano_lbs(mano, betas, full_pose):
    """lbs(), code/src/utils/external/lbs.py:139-251 with MANO.forward (body_models.py:601-685:
    full_pose += pose_mean).  mano: dict of torch tensors (v_template, shapedirs[778,3,10],
    posedirs[135,2334], J_regressor, parents, lbs_weights, pose_mean).
    Returns verts[B,778,3], joints[B,16,3], A[B,16,4,4], v_posed[B,778,3]."""
    B = full_pose.shape[0]
    dt = full_pose.dtype
    pose = full_pose + mano["pose_mean"]
    v_shaped = mano["v_template"] + torch.einsum("bl,mkl->bmk", betas, mano["shapedirs"])
    J = torch.einsum("bik,ji->bjk", v_shaped, mano["J_regressor"])
    rot = batch_rodrigues(pose.reshape(-1, 3)).view(B, -1, 3, 3)
    pose_feature = (rot[:, 1:] - torch.eye(3, dtype=dt)).reshape(B, -1)
    v_posed = v_shaped + torch.matmul(pose_feature, mano["posedirs"]).view(B, -1, 3)
    # batch_rigid_transform, lbs.py:345-399
    parents = mano["parents"]
    rel = J.clone()
    rel[:, 1:] = rel[:, 1:] - J[:, parents[1:]]
    tm = torch.cat([F.pad(rot.reshape(-1, 3, 3), [0, 0, 0, 1]),
                    F.pad(rel.reshape(-1, 3, 1), [0, 0, 0, 1], value=1.0)], 2).reshape(B, -1, 4, 4)
    chain = [tm[:, 0]]
    for i in range(1, parents.shape[0]):
        chain.append(torch.matmul(chain[int(parents[i])], tm[:, i]))
    T = torch.stack(chain, 1)
    posed_joints = T[:, :, :3, 3]
    Jh = F.pad(J.unsqueeze(-1), [0, 0, 0, 1])
    A = T - F.pad(torch.matmul(T, Jh), [3, 0, 0, 0, 0, 0, 0, 0])
    Tv = torch.matmul(mano["lbs_weights"][None].expand(B, -1, -1), A.view(B, -1, 16)).view(B, -1, 4, 4)
    vh = torch.cat([v_posed, torch.ones(B, v_posed.shape[1], 1, dtype=dt)], 2)
    verts = torch.matmul(Tv, vh.unsqueeze(-1))[:, :, :3, 0]
    return verts, posed_joints, A, v_posed


def mano_tensors(model: dict, dtype=torch.float32):
    """numpy MANO dict (hold_amd.synthetic.make_mano_model) -> tensors as SMPL.__init__ /
    MANO.__init__ register them (body_models.py:265-295, 549-561)."""
    t = lambda a: torch.as_tensor(np.asarray(a), dtype=dtype)
    parents = torch.as_tensor(model["kintree_table"][0].astype(np.int64)).clone()
    parents[0] = -1
    hand_mean = t(model["hands_mean"])
    return {
        "v_template": t(model["v_template"]),
        "shapedirs": t(model["shapedirs"][:, :, :10]),
        "posedirs": t(np.reshape(model["posedirs"], [-1, model["posedirs"].shape[-1]]).T),
        "J_regressor": t(model["J_regressor"]),
        "parents": parents,
        "lbs_weights": t(model["weights"]),
        "hand_mean": hand_mean,
        "pose_mean": torch.cat([torch.zeros(3, dtype=dtype), hand_mean]),
    }


def mano_server(mano, tfs_c_inv, scene_scale, transl, thetas, betas, absolute=False):
    """GenericServer.forward, code/src/model/mano/server.py:62-99 (MANO called with transl=0)."""
    verts, joints, A, v_posed = mano_lbs(mano, betas, thetas)
    tips = verts[:, torch.tensor([744, 320, 443, 554, 671])]
    joints = torch.cat([joints, tips], 1)
    s = scene_scale.view(-1, 1, 1)
    t = transl.view(-1, 1, 3)
    out = {"verts": verts * s + t * s, "jnts": joints * s + t * s, "v_posed": v_posed}
    tf = A.clone()
    tf[:, :, :3, :] = tf[:, :, :3, :] * s.view(-1, 1, 1, 1)
    tf[:, :, :3, 3] = tf[:, :, :3, 3] + t * s
    if not absolute:
        tf = torch.einsum("bnij,njk->bnik", tf, tfs_c_inv)
    out["tfs"] = tf
    return out


def mano_canonical(mano, betas):
    """canonical pose = -hand_mean, scale 1, transl 0 (server.py:11-17, 44-60); returns
    (verts_c[1,778,3], tfs_c_inv[16,4,4]).  betas: [10] mean shape of the sequence."""
    dt = mano["v_template"].dtype
    thetas = torch.zeros(1, 48, dtype=dt)
    thetas[0, 3:] = -mano["hand_mean"]
    out = mano_server(mano, None, torch.ones(1, dtype=dt), torch.zeros(1, 3, dtype=dt), thetas,
                      betas.view(1, 10).to(dt), absolute=True)
    return out["verts"], out["tfs"].squeeze(0).inverse()


def axis_angle_to_matrix(aa):
    """common/rot.py:105-138,777-805 (axis-angle -> quaternion -> matrix)."""
    angles = torch.norm(aa, p=2, dim=-1, keepdim=True)
    half = angles * 0.5
    small = angles.abs() < 1e-6
    soa = torch.where(small, 0.5 - (angles * angles) / 48, torch.sin(half) / torch.where(small, torch.ones_like(angles), angles))
    q = torch.cat([torch.cos(half), aa * soa], -1)
    r, i, j, k = torch.unbind(q, -1)
    two_s = 2.0 / (q * q).sum(-1)
    o = torch.stack((1 - two_s * (j * j + k * k), two_s * (i * j - k * r), two_s * (i * k + j * r),
                     two_s * (i * j + k * r), 1 - two_s * (i * i + k * k), two_s * (j * k - i * r),
                     two_s * (i * k - j * r), two_s * (j * k + i * r), 1 - two_s * (i * i + j * j)), -1)
    return o.reshape(q.shape[:-1] + (3, 3))


def object_server(rot, trans, scene_scale, obj_scale, denorm_mat, v3d_cano=None):
    """ObjectModel.forward, code/src/model/obj/object_model.py:29-70."""
    B = rot.shape[0]
    dt = rot.dtype
    tf = torch.eye(4, dtype=dt).repeat(B, 1, 1)
    tf[:, :3, :3] = axis_angle_to_matrix(rot)
    tf[:, :3, 3] = trans
    sm = torch.eye(4, dtype=dt).repeat(B, 1, 1) * scene_scale[:, None, None]
    sm[:, 3, 3] = 1
    om = torch.eye(4, dtype=dt).repeat(B, 1, 1) * obj_scale
    om[:, 3, 3] = 1
    tf = torch.matmul(torch.matmul(torch.matmul(sm, tf), om), denorm_mat[None].repeat(B, 1, 1))
    out = {"obj_tfs": tf}
    if v3d_cano is not None:
        vp = torch.cat([v3d_cano, torch.ones(v3d_cano.shape[0], 1, dtype=dt)], 1)[None].repeat(B, 1, 1)
        v = torch.bmm(tf, vp.permute(0, 2, 1)).permute(0, 2, 1)
        out["verts"] = v[:, :, :3] / v[:, :, 3:4]
    return out


# ------------------------------------------------------------------------------------------
# deformers
# ------------------------------------------------------------------------------------------


def knn_points(p1, p2, K):
    """pytorch3d 0.7.4 ops.knn_points: squared L2 of the K nearest, ascending."""
    d = ((p1[:, :, None, :] - p2[:, None, :, :]) ** 2).sum(-1)
    return torch.topk(d, K, dim=-1, largest=False, sorted=True)


def query_skinning_weights(pts, verts, skin_weights, K=15):
    """KNNDeformer.query_skinning_weights_multi, code/src/model/mano/deformer.py:84-105.
    pts [B,P,3], verts [B,778,3], skin_weights [778,16] -> w [B,P,16] (detached), outlier[B,P]."""
    out_w, out_o = [], []
    for c in range(0, pts.shape[1], 8192):
        p = pts[:, c:c + 8192]
        d, idx = knn_points(p.detach(), verts.detach(), K)
        d = torch.clamp(d, max=4)
        conf = torch.exp(-d)
        conf = conf / conf.sum(-1, keepdim=True)
        wk = skin_weights[idx]  # [B,P,K,16]
        out_w.append((wk * conf.unsqueeze(-1)).sum(2).detach())
        out_o.append(torch.sqrt(d).min(2).values > 0.1)
    return torch.cat(out_w, 1), torch.cat(out_o, 1)


def skinning(x, w, tfs, inverse=False):
    """code/src/model/mano/deformer.py:145-170."""
    xh = F.pad(x, (0, 1), value=1.0)
    if inverse:
        wtf = torch.einsum("bpn,bnij->bpij", w, tfs)
        xh = torch.einsum("bpij,bpj->bpi", wtf.inverse(), xh)
    else:
        xh = torch.einsum("bpn,bnij,bpj->bpi", w, tfs, xh)
    return xh[:, :, :3]


def rigid(x, tfs, inverse=False):
    """ObjectDeformer.forward, code/src/model/obj/deformer.py:10-41 (tfs [B,4,4])."""
    T = torch.inverse(tfs) if inverse else tfs
    xp = torch.cat([x, torch.ones_like(x[:, :, :1])], -1).permute(0, 2, 1)
    return torch.bmm(T, xp).permute(0, 2, 1)[:, :, :3]


# ------------------------------------------------------------------------------------------
# compositor
# ------------------------------------------------------------------------------------------


def density2weight(density, z_vals, z_max):
    """code/src/engine/volsdf_utils.py:220-251."""
    dists = torch.cat([z_vals[:, 1:] - z_vals[:, :-1], z_max.unsqueeze(-1) - z_vals[:, -1:]], -1)
    fe = dists * density
    alpha = 1 - torch.exp(-fe)
    sfe = torch.cat([torch.zeros(dists.shape[0], 1, dtype=density.dtype), fe], -1)
    trans = torch.exp(-torch.cumsum(sfe, -1))
    return alpha * trans[:, :-1], trans[:, -1]


def volumetric_render(f, is_training=False):
    """code/src/hold/hold_utils.py:243-271."""
    w, bgw = density2weight(f["density"].reshape(-1, f["z_vals"].shape[1]), f["z_vals"], f["z_max"])
    integ = lambda c: (c * w[:, :, None]).sum(1)
    out = {
        "fg_rgb": integ(f["color"]), "fg_weights": w,
        "mask_prob": torch.clamp(w.sum(1, keepdim=True), 0, 1),
        "normal": integ(f["normal"]), "depth": integ(f["z_vals"][:, :, None]),
        "fg_semantics": integ(f["semantics"]), "bg_weights": bgw,
    }
    if not is_training:
        out["fg_rgb.vis"] = out["fg_rgb"] + bgw[:, None]
    return out


def merge_factors(flist, stable=False):
    """code/src/hold/hold_utils.py:76-121 (incl. the CVPR off-by-one trim at :112-118).
    The reference calls torch.sort without ``stable``: the order of EQUAL z values (both nodes always
    contain near=0 and the sphere exit, and in eval mode share the uniform 'extra' samples) is
    backend-defined.  ``stable=True`` pins it to "earlier node first", the order the HIP merge uses."""
    comp = {k: torch.cat([f[k] for f in flist], 1) for k in flist[0]}
    z, idx = torch.sort(comp["z_vals"], dim=1, stable=stable)
    out = {"z_vals": z}
    for k, v in comp.items():
        if k != "z_vals":
            out[k] = torch.gather(v, 1, idx[:, :, None].repeat(1, 1, v.shape[-1]))
    n = len(flist)
    out = {k: v[:, (n - 1): -n] for k, v in out.items()}
    out["z_max"] = z[:, -n]
    return out


# ------------------------------------------------------------------------------------------
# background
# ------------------------------------------------------------------------------------------


def depth2pts_outside(ray_o, ray_d, depth, R):
    """code/src/model/renderables/background.py:102-135."""
    o_dot_d = torch.sum(ray_d * ray_o, -1)
    under = o_dot_d ** 2 - ((ray_o ** 2).sum(-1) - R ** 2)
    d_sphere = torch.sqrt(under) - o_dot_d
    p_sphere = ray_o + d_sphere.unsqueeze(-1) * ray_d
    p_mid = ray_o - o_dot_d.unsqueeze(-1) * ray_d
    p_mid_norm = torch.norm(p_mid, dim=-1)
    rot_axis = torch.cross(ray_o, p_sphere, dim=-1)
    rot_axis = rot_axis / torch.norm(rot_axis, dim=-1, keepdim=True)
    phi = torch.asin(p_mid_norm / R)
    theta = torch.asin(p_mid_norm * depth)
    ra = (phi - theta).unsqueeze(-1)
    p_new = (p_sphere * torch.cos(ra) + torch.cross(rot_axis, p_sphere, dim=-1) * torch.sin(ra)
             + rot_axis * torch.sum(rot_axis * p_sphere, -1, keepdim=True) * (1.0 - torch.cos(ra)))
    p_new = p_new / torch.norm(p_new, dim=-1, keepdim=True)
    return torch.cat((p_new, depth.unsqueeze(-1)), -1)


def background_rgb(sd, ray_dirs, cam_loc, z_bg, frame_latent_per_ray, R):
    """Background.bg_rendering, code/src/model/renderables/background.py:56-100,137-165.
    z_bg [N,32] ascending inverse depths; frame_latent_per_ray [N,32]."""
    N, S = z_bg.shape
    z = torch.flip(z_bg, dims=[-1])
    dirs = ray_dirs.unsqueeze(1).repeat(1, S, 1)
    locs = cam_loc.unsqueeze(1).repeat(1, S, 1)
    pts = depth2pts_outside(locs, dirs, z, R).reshape(-1, 4)
    lat = frame_latent_per_ray.unsqueeze(1).repeat(1, S, 1).reshape(-1, 32)
    out = implicit_net(sd, "background.bg_implicit_network", pts, lat, 10)
    sdf, feat = out[:, :1], out[:, 1:]
    rgb = rendering_net_bg(sd, "background.bg_rendering_network", dirs.reshape(-1, 3), lat, feat).reshape(N, S, 3)
    dens = sdf.abs().reshape(N, S)
    d = torch.cat([z[:, :-1] - z[:, 1:], torch.full((N, 1), 1e10, dtype=z.dtype)], -1)
    fe = d * dens
    sfe = torch.cat([torch.zeros(N, 1, dtype=z.dtype), fe[:, :-1]], -1)
    w = (1 - torch.exp(-fe)) * torch.exp(-torch.cumsum(sfe, -1))
    return (w.unsqueeze(-1) * rgb).sum(1)


# ------------------------------------------------------------------------------------------
# cameras
# ------------------------------------------------------------------------------------------


def get_camera_params(uv, pose, intrinsics):
    """code/src/datasets/utils.py:230-282 (matrix-pose branch). uv [B,P,2] -> dirs [B,P,3], cam_loc [B,3]."""
    cam_loc = pose[:, :3, 3]
    fx, fy = intrinsics[:, 0, 0, None], intrinsics[:, 1, 1, None]
    cx, cy, sk = intrinsics[:, 0, 2, None], intrinsics[:, 1, 2, None], intrinsics[:, 0, 1, None]
    x, y = uv[:, :, 0], uv[:, :, 1]
    z = torch.ones_like(x)
    xl = (x - cx + cy * sk / fy - sk * y / fy) / fx * z
    yl = (y - cy) / fy * z
    pc = torch.stack((xl, yl, z, torch.ones_like(z)), -1).permute(0, 2, 1)
    world = torch.bmm(pose, pc).permute(0, 2, 1)[:, :, :3]
    return F.normalize(world - cam_loc[:, None, :], dim=2), cam_loc


# ------------------------------------------------------------------------------------------
# whole forward (HOLDNet.forward, code/src/hold/hold_net.py:53-134), functional
# ------------------------------------------------------------------------------------------


class OracleScene:
    """Holds constant tensors derived from the scene + MANO model (what the reference
    computes in the MANOServer / MANODeformer / ObjectServer constructors)."""

    def __init__(self, scene, mano_models: dict, dtype=torch.float32, N_samples=64):
        self.dtype = dtype
        # ray_sampler.N_samples of the config (general.yaml:71): 64 shipped; BASELINE.json configs[0] uses 32, configs[4] 128
        self.N_samples = N_samples
        self.R = float(scene["scene_bounding_sphere"])
        self.nodes = list(scene["entities"].keys())
        self.mano, self.verts_c, self.tfs_c_inv, self.skin_w = {}, {}, {}, {}
        for n in self.nodes:
            if n == "object":
                ent = scene["entities"][n]
                self.obj_scale = torch.as_tensor(np.array([ent["obj_scale"]]), dtype=dtype)
                self.denorm = torch.inverse(torch.as_tensor(ent["norm_mat"], dtype=dtype))
                continue
            m = mano_tensors(mano_models[n], dtype)
            betas = torch.as_tensor(scene["entities"][n]["mean_shape"], dtype=dtype)
            vc, tci = mano_canonical(m, betas)
            self.mano[n], self.verts_c[n], self.tfs_c_inv[n], self.skin_w[n] = m, vc, tci, m["lbs_weights"]


def node_forward(osc: OracleScene, sd, node, inp, ray_dirs, cam_loc, is_training, rng=None,
                 z_vals_override=None, current_epoch=0, barf_alpha_iter=None, extras=None):
    """Node.forward (code/src/model/renderables/node.py:49-87) + sample_points
    (mano_node.py:71-124, object_node.py:57-110) for one node.  ray_dirs/cam_loc: [B*P,3]."""
    dt = osc.dtype
    B = inp["idx"].shape[0]
    Ntot = ray_dirs.shape[0]
    scale = inp[f"{node}.params"][:, 0]
    is_obj = node == "object"
    class_id = {"object": 1, "right": 2, "left": 3}[node]
    if is_obj:
        so = object_server(inp["object.global_orient"], inp["object.transl"], scale, osc.obj_scale, osc.denorm)
        tfs = so["obj_tfs"]  # [B,4,4]
        cond_pose = torch.zeros(B, 0, dtype=dt)
        embed_w = None
        if is_training and barf_alpha_iter is not None:
            embed_w = barf_weights(barf_alpha_iter, 6, 3)
        inv = lambda x: rigid(x, tfs, inverse=True)
        verts = None
    else:
        full_pose = torch.cat([inp[f"{node}.global_orient"], inp[f"{node}.pose"]], 1)
        so = mano_server(osc.mano[node], osc.tfs_c_inv[node], scale, inp[f"{node}.transl"], full_pose,
                         inp[f"{node}.betas"])
        tfs, verts = so["tfs"], so["verts"]
        cond_pose = full_pose[:, 3:] / np.pi
        if is_training and current_epoch < 20:
            cond_pose = full_pose[:, 3:] * 0.0
        embed_w = None

        def inv(x):
            w, _ = query_skinning_weights(x, verts, osc.skin_w[node])
            return skinning(x, w, tfs, inverse=True)

    prefix = f"nodes.{node}.implicit_network"

    def sdf_only(pts_flat):
        xc = inv(pts_flat.view(B, -1, 3)).reshape(-1, 3)
        cond = None if is_obj else cond_pose[:, None, :].expand(B, xc.shape[0] // B, 45).reshape(-1, 45)
        return implicit_net(sd, prefix, xc, cond, 6, embed_w, zero_cond=not is_obj)[:, 0]

    beta0 = get_beta(sd, node).detach()
    far = sphere_far(cam_loc, ray_dirs, osc.R)
    z0 = uniform_z(torch.zeros(Ntot, 1, dtype=dt), far, 128, rng["t_uniform"] if is_training else None)
    iters = -1
    if z_vals_override is None:
        with torch.no_grad():
            z_vals, iters = error_bound_sample(z0, sdf_only, cam_loc, ray_dirs, beta0, osc.R, is_training, rng,
                                               N_samples=getattr(osc, "N_samples", 64),
                                               trace=(extras.setdefault("trace", []) if extras is not None else None))
    else:
        z_vals = z_vals_override
    S = z_vals.shape[1]
    pts = (cam_loc.unsqueeze(1) + z_vals.unsqueeze(2) * ray_dirs.unsqueeze(1)).view(B, -1, 3)
    x_c = inv(pts)  # [B,P*S,3]
    xc_flat = x_c.reshape(-1, 3)
    if not xc_flat.requires_grad:
        xc_flat.requires_grad_(True)
    # extract_features, code/src/engine/volsdf_utils.py:51-105
    if is_obj:
        Jm = tfs[:, None, :3, :3].expand(B, xc_flat.shape[0] // B, 3, 3).reshape(-1, 3, 3)
        cond = None
    else:
        w_c, _ = query_skinning_weights(x_c.detach(), osc.verts_c[node].expand(B, -1, -1), osc.skin_w[node])
        Jm = torch.einsum("bpn,bnij->bpij", w_c, tfs)[:, :, :3, :3].reshape(-1, 3, 3)
        cond = cond_pose[:, None, :].expand(B, xc_flat.shape[0] // B, 45).reshape(-1, 45)
    out = implicit_net(sd, prefix, xc_flat, cond, 6, embed_w, zero_cond=not is_obj)
    sdf, feat = out[:, :1], out[:, 1:]
    g = torch.autograd.grad(sdf, xc_flat, torch.ones_like(sdf), create_graph=is_training, retain_graph=True)[0]
    normals = F.normalize(torch.einsum("bi,bij->bj", g, Jm.inverse()), dim=1, eps=1e-6)
    if is_obj:
        tc = inp["object.time_code"][:, None, :].expand(B, xc_flat.shape[0] // B, 32).reshape(-1, 32)
        feat_in = torch.cat([feat, tc], -1)
        bp = torch.zeros(xc_flat.shape[0], 0, dtype=dt)
    else:
        feat_in = feat
        bp = cond_pose[:, None, :].expand(B, xc_flat.shape[0] // B, 45).reshape(-1, 45)
    rgb = rendering_net_pose(sd, f"nodes.{node}.rendering_network", xc_flat, normals, bp, feat_in)
    density = laplace_density(sdf, get_beta(sd, node)).view(-1, S, 1)
    sem = torch.zeros(Ntot, S, 4, dtype=dt)
    sem[:, :, class_id] = 1.0
    factors = {"color": rgb.reshape(-1, S, 3), "normal": normals.reshape(-1, S, 3), "density": density,
               "semantics": sem, "z_vals": z_vals}
    if extras is not None:
        extras.update(dict(x_c=xc_flat, sdf=sdf, feat=feat, grad=g, tfs=tfs, verts=verts, iters=iters, server=so))
    return factors


def holdnet_forward(osc: OracleScene, sd, inp, is_training=False, rng=None, z_override=None,
                    current_epoch=0, barf_alpha_iter=None, extras=None, stable_merge=False):
    """HOLDNet.forward (code/src/hold/hold_net.py:53-134) without the kaolin loss targets."""
    dt = osc.dtype
    ray_dirs, cam = get_camera_params(inp["uv"], inp["extrinsics"], inp["intrinsics"])
    B, P, _ = ray_dirs.shape
    cam_loc = cam.unsqueeze(1).repeat(1, P, 1).reshape(-1, 3)
    ray_dirs = ray_dirs.reshape(-1, 3)
    if "object" in osc.nodes:
        inp = dict(inp)
        inp["object.time_code"] = sd["nodes.object.frame_latent_encoder.weight"][inp["idx"]]
    fdict = {}
    for node in osc.nodes:
        ex = None if extras is None else extras.setdefault(node, {})
        fdict[node] = node_forward(osc, sd, node, inp, ray_dirs, cam_loc, is_training,
                                   None if rng is None else rng[node],
                                   None if z_override is None else z_override[node],
                                   current_epoch, barf_alpha_iter, ex)
    out = {}
    comp = merge_factors(list(fdict.values()), stable=stable_merge)
    for f in fdict.values():
        f["z_max"] = f["z_vals"][:, -1]
    out.update(volumetric_render(comp, is_training))
    for node, f in fdict.items():
        out.update({f"{node}.{k}": v for k, v in volumetric_render(f, is_training).items()})
    z_bg = uniform_z(torch.zeros(B * P, 1, dtype=dt), torch.ones(B * P, 1, dtype=dt), 32,
                     rng["bg_t"] if is_training else None) * (1.0 / osc.R)
    lat = sd["background.frame_latent_encoder.weight"][inp["idx"]]
    lat_ray = lat[:, None, :].expand(B, P, 32).reshape(-1, 32)
    bg_only = background_rgb(sd, ray_dirs, cam_loc, z_bg, lat_ray, osc.R)
    out["bg_rgb_only"] = bg_only
    out["rgb"] = out["fg_rgb"] + out["bg_weights"].unsqueeze(-1) * bg_only
    bg_sem = torch.zeros(B * P, 4, dtype=dt)
    bg_sem[:, 0] = 1.0
    out["semantics"] = out["fg_semantics"] + out["bg_weights"].unsqueeze(-1) * bg_sem
    if not is_training:
        out["instance_map"] = torch.argmax(out["semantics"], dim=1)
    out["bg_z_vals"] = z_bg
    out["ray_dirs"], out["cam_loc"] = ray_dirs, cam_loc
    for node, f in fdict.items():
        out[f"{node}.z_vals"] = f["z_vals"]
    return out
