"""GPU parity of the pose-refinement kernels (soft silhouette, K=1 neighbour, fitting loop) against the torch
restatement in oracle/fitting_oracle.py (pytorch3d itself is unavailable: parity of that dependency is unpinned)."""
import numpy as np
import pytest
import torch

from parity_common import hip_net, setup, syn
from oracle import fitting_oracle as fo

pytestmark = pytest.mark.gpu


def _scene_verts(n_frames=2):
    sc, sd_np, sd, osc = setup(n_frames=4)
    net = hip_net(sc, sd_np)
    node = net.nodes["right"]
    idx = torch.arange(n_frames, device="cuda")
    p = node.params(idx)
    with torch.no_grad():
        so = node.server(torch.full((n_frames,), 1.0, device="cuda"), p["right.transl"], p["right.full_pose"],
                         p["right.betas"])
    verts = so["verts"].clone()
    verts[..., :2] -= verts[..., :2].mean(dim=1, keepdim=True)  # centred on the optical axis
    verts[..., 2] += 0.45  # in front of the camera
    faces = torch.as_tensor(node.server.faces.astype(np.int64), device="cuda")
    return net, verts, faces


@pytest.mark.parametrize("sigma", [1e-4, 1e-6])
def test_silhouette_forward(sigma):
    from hold_amd import fitting as ft
    net, verts, faces = _scene_verts()
    H = W = 64
    fx = fy = 300.0
    cx = cy = 32.0
    blur = np.log(1.0 / 1e-4 - 1.0) * sigma
    vs, fs = ft.seal_mano_mesh(verts, faces, True)
    m = ft.soft_silhouette(vs, fs, fx, fy, cx, cy, H, W, sigma, blur).cpu()
    vo, fo_ = fo.seal_mano_mesh(verts.cpu(), faces.cpu(), True)
    ref = fo.soft_silhouette(vo, fo_, fx, fy, cx, cy, H, W, sigma, blur)
    assert 0.03 < float(ref.mean()) < 0.9
    diff = (m - ref).abs()
    if sigma >= 1e-4:
        assert float(diff.max()) < 2e-3
    else:  # sigma 1e-6: the 0.15-pixel edge band amplifies fp32 differences of d by 1e6
        assert float(diff.mean()) < 2e-3 and float((diff > 0.05).float().mean()) < 0.01


def test_silhouette_backward():
    from hold_amd import fitting as ft
    net, verts, faces = _scene_verts(1)
    H = W = 48
    fx = fy = 220.0
    cx = cy = 24.0
    sigma = 1e-4
    blur = np.log(1.0 / 1e-4 - 1.0) * sigma
    g = torch.Generator().manual_seed(0)
    wgt = torch.rand(1, H, W, generator=g)
    v = verts.clone().requires_grad_(True)
    vs, fs = ft.seal_mano_mesh(v, faces, True)
    (ft.soft_silhouette(vs, fs, fx, fy, cx, cy, H, W, sigma, blur) * wgt.cuda()).sum().backward()
    vc = verts.cpu().clone().requires_grad_(True)
    vo, fo_ = fo.seal_mano_mesh(vc, faces.cpu(), True)
    (fo.soft_silhouette(vo, fo_, fx, fy, cx, cy, H, W, sigma, blur) * wgt).sum().backward()
    rel = float((v.grad.cpu() - vc.grad).norm() / vc.grad.norm())
    assert rel < 2e-2, rel


def _perturbed(verts, rel, seed):
    g = torch.Generator().manual_seed(seed)
    return verts * (1.0 + rel * (2.0 * torch.rand(verts.shape, generator=g, dtype=verts.dtype) - 1.0))


def test_silhouette_at_the_reference_sigma_under_a_conditioning_aware_bound():
    """at the reference's sigma = 1e-6 (fitting/utils.py:101-118) the silhouette is a step function of the
    signed squared NDC distance d: alpha moves from 0 to 1 while d crosses a band 1e-5 wide, so the fp32 rounding of the NDC
    arithmetic (relative 1e-7 of coordinates of order 1, i.e. 1e-7 in d) decides alpha at the pixels whose centre lies within
    that band of an edge -- for ANY fp32 implementation, pytorch3d's included.  As for the sampler's inverse CDF
    (oracle.inv_cdf_conditioning), the test therefore measures how far the fp64 oracle's alpha moves at each pixel when the
    vertices move by fp32-sized noise (relative 2e-7, eight draws) and holds the HIP kernel, forward AND backward, to
        |alpha_hip - alpha_fp64| <= 1e-4 + 4 x spread(pixel)
    -- 1e-4 at every well-conditioned pixel (spread < 1e-5: all but the edge band) -- and the vertex gradient of a weighted sum
    of alpha to 1e-3 of its norm + 4 x the spread of the fp64 gradient under the same noise."""
    from hold_amd import fitting as ft
    net, verts, faces = _scene_verts(1)
    H = W = 64
    fx = fy = 300.0
    cx = cy = 32.0
    sigma = 1e-6
    blur = np.log(1.0 / 1e-4 - 1.0) * sigma
    g = torch.Generator().manual_seed(0)
    wgt = torch.rand(1, H, W, generator=g, dtype=torch.float64)
    v = verts.clone().requires_grad_(True)
    vs, fs = ft.seal_mano_mesh(v, faces, True)
    m = ft.soft_silhouette(vs, fs, fx, fy, cx, cy, H, W, sigma, blur)
    (m * wgt.float().cuda()).sum().backward()
    m = m.detach().cpu().double()
    gh = v.grad.cpu().double()

    def oracle(vin):
        vc = vin.clone().requires_grad_(True)
        vo, fo_ = fo.seal_mano_mesh(vc, faces.cpu(), True)
        a = fo.soft_silhouette(vo, fo_, fx, fy, cx, cy, H, W, sigma, blur)
        (a * wgt).sum().backward()
        return a.detach(), vc.grad.detach()

    v64 = verts.cpu().double()
    ref, gref = oracle(v64)
    assert 0.03 < float(ref.mean()) < 0.9
    spread = torch.zeros_like(ref)
    gspread = 0.0
    for k in range(8):
        a, gk = oracle(_perturbed(v64, 2e-7, k))
        spread = torch.maximum(spread, (a - ref).abs())
        gspread = max(gspread, float((gk - gref).norm()))
    err = (m - ref).abs()
    well = spread < 1e-5
    print(f"sigma 1e-6: {float((~well).double().mean()) * 100:.2f} % of the pixels are edge-band pixels; max err well-conditioned "
          f"{float(err[well].max()):.2e}, edge band {float(err[~well].max()) if (~well).any() else 0.0:.2e} (spread up to {float(spread.max()):.2e}); "
          f"gradient: |g_hip - g_64| / |g_64| = {float((gh - gref).norm() / gref.norm()):.2e}, fp64 spread / |g_64| = {gspread / float(gref.norm()):.2e}")
    assert float(well.double().mean()) > 0.9
    assert bool((err <= 1e-4 + 4.0 * spread).all()), float((err - 4.0 * spread).max())
    assert float((gh - gref).norm()) <= 1e-3 * float(gref.norm()) + 4.0 * gspread


def test_faces_straddling_the_image_plane_are_refused():
    """hold_silhouette_fwd drops a face with a vertex behind the camera as a whole, pytorch3d per pixel (oracle cull='pixel'):
    check_faces_per_pixel -- the guard Model.forward calls once per fit -- must refuse such a configuration"""
    from hold_amd import fitting as ft
    tri = torch.tensor([[[-0.05, -0.05, 0.5], [0.05, -0.05, 0.5], [0.0, 0.4, -0.1]]], device="cuda")
    faces = torch.tensor([[0, 1, 2]], device="cuda")
    with pytest.raises(NotImplementedError, match="straddles"):
        ft.check_faces_per_pixel(tri, faces, 300.0, 300.0, 32.0, 32.0, 64, 64)
    front = tri.clone()
    front[..., 2] = 0.5
    assert ft.check_faces_per_pixel(front, faces, 300.0, 300.0, 32.0, 32.0, 64, 64) >= 1


def test_knn1():
    from hold_amd import fitting as ft
    g = torch.Generator().manual_seed(1)
    q = torch.randn(3, 70, 3, generator=g).cuda().requires_grad_(True)
    t = torch.randn(3, 900, 3, generator=g).cuda().requires_grad_(True)
    d = ft.knn1_sqdist(q, t)
    ref = ((q[:, :, None] - t[:, None]) ** 2).sum(-1).min(-1).values
    assert float((d - ref).abs().max()) < 1e-5
    w = torch.rand(3, 70, generator=g).cuda()
    gq, gt = torch.autograd.grad((d * w).sum(), (q, t))
    rq, rt = torch.autograd.grad((ref * w).sum(), (q, t))
    assert float((gq - rq).abs().max()) < 1e-5 and float((gt - rt).abs().max()) < 1e-5


def test_fitting_loop_reduces_loss():
    """Model.fit-style refinement: perturb hand translation / object pose, fit against masks rendered from the truth."""
    from hold_amd import fitting as ft
    sc, sd_np, sd, osc = setup(n_frames=4)
    net = hip_net(sc, sd_np)
    B = 3
    dev = torch.device("cuda")
    hand, obj = net.nodes["right"], net.nodes["object"]
    idx = torch.arange(B, device=dev)
    hp, op = hand.params(idx), obj.params(idx)
    params = {"scene_scale": torch.tensor([1.0], device=dev), "right.global_orient": hp["right.global_orient"].detach(),
              "right.pose": hp["right.pose"].detach(), "right.betas": hp["right.betas"][:1].detach(),
              "right.transl": hp["right.transl"].detach(), "object.global_orient": op["object.global_orient"].detach(),
              "object.transl": op["object.transl"].detach()}
    w2c = torch.eye(4, device=dev).repeat(B, 1, 1)
    w2c[:, 2, 3] = 0.9
    K = torch.tensor([[260.0, 0, 40.0], [0, 260.0, 40.0], [0, 0, 1]], device=dev)
    hand_faces = torch.as_tensor(hand.server.faces.astype(np.int64), device=dev)
    # object mesh: a small lat-long sphere (radius 0.07) replaces the synthetic point cloud
    nlat, nlon = 12, 16
    th = torch.linspace(0.15, np.pi - 0.15, nlat)
    ph = torch.linspace(0, 2 * np.pi, nlon + 1)[:-1]
    sv = torch.stack([torch.sin(th)[:, None] * torch.cos(ph)[None], torch.sin(th)[:, None] * torch.sin(ph)[None],
                      torch.cos(th)[:, None].expand(nlat, nlon)], -1).reshape(-1, 3) * 0.07
    obj.server.object_model.v3d_cano = sv.to(dev)
    fl = []
    for a in range(nlat - 1):
        for b in range(nlon):
            i0, i1 = a * nlon + b, a * nlon + (b + 1) % nlon
            fl += [[i0, i1, i0 + nlon], [i1, i1 + nlon, i0 + nlon]]
    obj_faces = torch.tensor(fl, device=dev)
    contact_idx = torch.arange(700, 778, device=dev)
    gt = ft.FittingModel(hand.server, obj.server, hand_faces, obj_faces, params, w2c, K, (80, 80), None, contact_idx)
    with torch.no_grad():
        o = gt.fwd_params()
    targets = {"right": (o["right.mask"] > 0.5).float(), "object": (o["object.mask"] > 0.5).float()}
    p2 = dict(params)
    p2["right.transl"] = params["right.transl"] + torch.tensor([0.01, -0.008, 0.0], device=dev)
    p2["object.transl"] = params["object.transl"] + torch.tensor([-0.02, 0.015, 0.0], device=dev)
    m = ft.FittingModel(hand.server, obj.server, hand_faces, obj_faces, p2, w2c, K, (80, 80), targets, contact_idx)
    hist = m.fit(num_iterations=40)
    assert hist[-1] < 0.7 * hist[0], (hist[0], hist[-1])
    assert all(np.isfinite(hist))


def test_fitting_losses_match_reference_golden():
    """hold_amd.fitting.loss_fn_h / loss_fn_ih (HIP K=1 neighbour search + its backward inside) on the golden inputs
    against the REFERENCE's outputs and gradients (tests/golden/fitting_losses.npz)."""
    import os
    from fitting_loss_cases import run_single_hand, run_two_hand
    from hold_amd import fitting as ft
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "fitting_losses.npz"))
    assert run_two_hand(g, ft.loss_fn_ih, "cuda") < 2e-5
    assert run_single_hand(g, ft.loss_fn_h, "cuda") < 2e-5


def test_reference_model_and_optimize_batch_surface():
    """Model(servers, scene_scale, obj_scale, param_dict, device, target_masks, w2c, K, fnames, faces) and
    optimize_batch(batch_idx, args, pbar, out, device, ...) with the reference's signatures (fitting/model.py:30-200,
    fitting/fitting.py:22-76): same first-iteration loss as the explicit single-hand FittingModel, requires_grad
    pattern of optimize_batch, and a loss that goes down."""
    import types
    from hold_amd import fitting as ft
    sc, sd_np, sd, osc = setup(n_frames=6)
    net = hip_net(sc, sd_np)
    dev = torch.device("cuda")
    hand, obj = net.nodes["right"], net.nodes["object"]
    n = sc["n_frames"]
    nlat, nlon = 12, 16
    th = torch.linspace(0.15, np.pi - 0.15, nlat)
    ph = torch.linspace(0, 2 * np.pi, nlon + 1)[:-1]
    sv = torch.stack([torch.sin(th)[:, None] * torch.cos(ph)[None], torch.sin(th)[:, None] * torch.sin(ph)[None],
                      torch.cos(th)[:, None].expand(nlat, nlon)], -1).reshape(-1, 3) * 0.07
    obj.server.object_model.v3d_cano = sv.to(dev)
    fl = []
    for a in range(nlat - 1):
        for b in range(nlon):
            i0, i1 = a * nlon + b, a * nlon + (b + 1) % nlon
            fl += [[i0, i1, i0 + nlon], [i1, i1 + nlon, i0 + nlon]]
    faces = {"right": torch.as_tensor(hand.server.faces.astype(np.int64), device=dev), "object": torch.tensor(fl, device=dev)}
    pre = "model.nodes."
    pd = {pre + "right.params.global_orient.weight": hand.params.global_orient.weight.detach().clone(),
          pre + "right.params.pose.weight": hand.params.pose.weight.detach().clone(),
          pre + "right.params.betas.weight": hand.params.betas.weight.detach().clone(),
          pre + "right.params.transl.weight": hand.params.transl.weight.detach().clone(),
          pre + "object.params.global_orient.weight": obj.params.global_orient.weight.detach().clone(),
          pre + "object.params.transl.weight": obj.params.transl.weight.detach().clone()}
    w2c = torch.eye(4, device=dev)[None]
    w2c[:, 2, 3] = 0.9
    K = torch.tensor([[[260.0, 0, 40.0], [0, 260.0, 40.0], [0, 0, 1]]], device=dev)
    out = dict(servers={"right": hand.server, "object": obj.server}, faces=faces, K=K, w2c=w2c,
               scene_scale=torch.tensor([1.0]), param_dict=pd, fnames=[f"{i:04d}.png" for i in range(n)])
    # target masks = hard silhouettes of the un-perturbed parameters, coded with SEGM ids (object 50, right 150)
    batch_idx = [0, 2, 3]
    contact_idx = np.arange(700, 778)
    args = types.SimpleNamespace(iters=0, vis_every=50, itw=True, write_gif=False)
    m0 = ft.optimize_batch(batch_idx, args, None, out, dev, obj_scale=[1.0], masks=np.zeros((n, 80, 80), np.float32),
                           contact_idx=contact_idx)
    with torch.no_grad():
        o = m0.fwd_params()
    assert m0.imsize == (300, 300) and o["right.mask"].shape == (3, 300, 300) and hasattr(o, "search")
    masks = np.zeros((n, 300, 300), np.float32)
    hard_o, hard_h = (o["object.mask"] > 0.5).cpu().numpy(), (o["right.mask"] > 0.5).cpu().numpy()
    for j, i in enumerate(batch_idx):
        masks[i][hard_o[j]] = 50
        masks[i][hard_h[j]] = 150
    # requires_grad pattern of fitting.py:57-67
    rg = {k: p.requires_grad for k, p in m0.param_dict.items()}
    assert rg == {"right__global_orient": False, "right__pose": False, "right__betas": True, "right__transl": True,
                  "object__global_orient": True, "object__transl": True, "right__scene_scale": False,
                  "object__scene_scale": False} and m0.obj_scale.requires_grad
    # perturbed start
    pd2 = dict(pd)
    pd2[pre + "right.params.transl.weight"] = pd[pre + "right.params.transl.weight"] + torch.tensor([0.01, -0.008, 0.0], device=dev)
    pd2[pre + "object.params.transl.weight"] = pd[pre + "object.params.transl.weight"] + torch.tensor([-0.02, 0.015, 0.0], device=dev)
    K300 = K.clone()  # the recorded masks are already 300 x 300: intrinsics of that size (scaling_masks_K then applies k = 1)
    K300[:, :2] *= 300.0 / 80.0
    # first-iteration loss of the explicit single-hand model on the same inputs (obj_scale is still the untouched 1.0)
    idx = torch.tensor(batch_idx, device=dev)
    params = {"scene_scale": torch.tensor([1.0], device=dev), "right.global_orient": pd2[pre + "right.params.global_orient.weight"][idx],
              "right.pose": pd2[pre + "right.params.pose.weight"][idx], "right.betas": pd2[pre + "right.params.betas.weight"],
              "right.transl": pd2[pre + "right.params.transl.weight"][idx],
              "object.global_orient": pd2[pre + "object.params.global_orient.weight"][idx],
              "object.transl": pd2[pre + "object.params.transl.weight"][idx]}
    tm = torch.as_tensor(masks[batch_idx]).to(dev)
    ref = ft.FittingModel(hand.server, obj.server, faces["right"], faces["object"], params, w2c.repeat(3, 1, 1), K300[0],
                          (300, 300), ft.construct_targets(tm), torch.as_tensor(contact_idx, device=dev))
    with torch.no_grad():
        l_ref = float(ref()["loss"])
    # 40 iterations through the reference-signature entry point
    args.iters = 40
    m = ft.optimize_batch(batch_idx, args, None, dict(out, param_dict=pd2, K=K300), dev, obj_scale=[1.0], masks=masks,
                          contact_idx=contact_idx, freeze_shape=True)
    assert len(m.history) == 40 and all(np.isfinite(m.history)) and m.history[-1] < 0.7 * m.history[0]
    assert m.history[0] == pytest.approx(l_ref, rel=1e-5)


def test_faces_per_pixel_cap_is_detected():
    """pytorch3d keeps the 100 nearest faces per pixel (fitting/utils.py:107); the HIP rasteriser multiplies over all
    of them, which is the same thing while no pixel sees more than 100: the count kernel equals the oracle's count on
    the sealed hand, and a stack of 150 coincident triangles trips the host-side check."""
    from hold_amd import fitting as ft
    net, verts, faces = _scene_verts(2)
    vs, fs = ft.seal_mano_mesh(verts, faces, True)
    H = W = 64
    fx = fy = 180.0
    cx = cy = 32.0
    k = ft.max_faces_per_pixel(vs, fs, fx, fy, cx, cy, H, W)
    _, cnt = fo.soft_silhouette(vs.cpu().double(), fs.cpu(), fx, fy, cx, cy, H, W, return_count=True)
    assert k == int(cnt.max()) and 2 <= k <= ft.FACES_PER_PIXEL
    assert ft.check_faces_per_pixel(vs, fs, fx, fy, cx, cy, H, W) == k
    tri = torch.tensor([[[-0.1, -0.1, 0.5], [0.1, -0.1, 0.5], [0.0, 0.1, 0.5]]], device="cuda")
    stack = torch.arange(150, device="cuda").repeat_interleave(3).view(150, 3) * 0 + torch.tensor([0, 1, 2], device="cuda")
    assert ft.max_faces_per_pixel(tri, stack, fx, fy, cx, cy, H, W) == 150
    with pytest.raises(NotImplementedError):
        ft.check_faces_per_pixel(tri, stack, fx, fy, cx, cy, H, W)
