"""GPU parity of the training-only side of the path: loss targets emitted by
HOLDNet.forward, the full Loss, the flat-bucket Adam step, ray generation, canonical meshing, the loss-target geometry."""
import math

import numpy as np
import pytest
import torch

from parity_common import hip_input, hip_net, ho, oracle_input, rel_err, setup

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    sc, sd_np, sd, osc = setup()
    return dict(sc=sc, sd_np=sd_np, sd=sd, osc=osc)


def _rng(sc, N, seed=5):
    g = torch.Generator().manual_seed(seed)
    rng = {"bg_t": torch.rand(N, 32, generator=g)}
    for i, n in enumerate(sc["entities"]):
        rng[n] = {"t_uniform": torch.rand(N, 128, generator=g), "u_final": torch.rand(N, 64, generator=g),
                  "perm": (lambda S, _s=i: torch.randperm(S, generator=torch.Generator().manual_seed(100 + _s)))}
    return rng


def _cuda_rng(rng):
    return {k: ({kk: (vv.cuda() if torch.is_tensor(vv) else vv) for kk, vv in v.items()} if isinstance(v, dict)
                else v.cuda()) for k, v in rng.items()}


# ---------------------------------------------------------------------------------------------- f-1
def test_raygen_matches_oracle(ctx):
    from hold_amd import kernels as K, synthetic as syn
    sc = ctx["sc"]
    uv = syn.make_uv(37, 29)
    b = syn.make_batch(sc, [0, 3], uv, 37, 29)
    t = {k: torch.from_numpy(v) for k, v in b.items()}
    rd, cl = ho.get_camera_params(t["uv"], t["extrinsics"], t["intrinsics"])
    dirs, cam = K.raygen(t["uv"].cuda(), t["extrinsics"].cuda(), t["intrinsics"].cuda())
    assert float((dirs.cpu() - rd.reshape(-1, 3)).abs().max()) < 2e-6
    assert torch.equal(cam.cpu(), cl[:, None, :].expand(-1, rd.shape[1], -1).reshape(-1, 3))
    assert float((dirs.norm(dim=1) - 1).abs().max()) < 1e-6


def test_inference_step_frame_at_once_equals_chunked(ctx):
    """inference_step (hold.py:169-208): the merged vis keys of one full-frame call == 512-pixel chunks, one D2H."""
    from hold_amd import synthetic as syn
    from hold_amd.train import inference_step
    sc = ctx["sc"]
    net = hip_net(sc, ctx["sd_np"])
    W = H = 40
    b = syn.make_batch(sc, [1], syn.make_uv(W, H), W, H)
    batch = {k: torch.from_numpy(v) for k, v in b.items()}
    batch["total_pixels"] = torch.tensor([W * H])
    batch["img_size"] = [torch.tensor([H]), torch.tensor([W])]
    full = inference_step(net, batch, chunk_rays=W * H)
    parts = inference_step(net, batch, chunk_rays=512)
    for k in ("rgb", "instance_map", "bg_rgb_only", "normal", "mask_prob", "right.fg_rgb.vis", "object.mask_prob"):
        assert full[k].device.type == "cpu" and full[k].shape[0] == W * H, k
    # the sampler's convergence flag is per call, so only sampler-independent structure is compared exactly
    assert float((full["rgb"] - parts["rgb"]).abs().mean()) < 2e-3
    assert hasattr(full, "search") and set(full.search("fg_rgb.vis").keys()) == {"fg_rgb.vis", "right.fg_rgb.vis", "object.fg_rgb.vis"}
    half = inference_step(net, batch, chunk_rays=W * H, render_downsample=2)
    assert half["rgb"].shape[0] == (W // 2) * (H // 2)


# ---------------------------------------------------------------------------------------------- f-2 geometry
def test_mesh_sdf_kernel_matches_oracle(ctx):
    """hold_mesh_sdf against the closed-form box SDF and against the fp64 oracle on the sealed, Loop-subdivided
    synthetic MANO (a closed embedded surface); bbox culling leaves the per-ray off / in-surface masks unchanged."""
    from hold_amd import fitting as ft, geometry as geo, synthetic as syn
    from oracle import geometry_oracle as go
    g = torch.Generator().manual_seed(0)
    h = (0.3, 0.2, 0.5)
    v, f = go.box_mesh(h)
    p = ((torch.rand(2, 3000, 3, generator=g) * 2 - 1) * 0.8)
    sd = geo.mesh_sdf(p.cuda(), v.float().cuda(), f.cuda())
    assert float((sd.cpu().double() - go.box_sdf(p.double(), h)).abs().max()) < 2e-6
    # > 32 768 points take the one-thread-per-point kernel, fewer the lanes-over-faces kernel: same values
    pl = ((torch.rand(1, 40000, 3, generator=g) * 2 - 1) * 0.8)
    sdl = geo.mesh_sdf(pl.cuda(), v.float().cuda(), f.cuda())
    assert float((sdl.cpu().double() - go.box_sdf(pl.double(), h)).abs().max()) < 2e-6
    assert torch.equal(geo.mesh_sdf(pl[:, :3000].contiguous().cuda(), v.float().cuda(), f.cuda()), sdl[:, :3000])
    m = syn.make_mano_model(True)
    verts = torch.tensor(m["v_template"], dtype=torch.float32)[None].cuda()
    faces = torch.tensor(m["f"]).cuda()
    vs, fs = ft.seal_mano_mesh(verts, faces, True)
    vd, fd = geo.subdivide_loop(vs[0], fs)
    assert vd.shape[0] == 3110 and fd.shape[0] == 6216  # mano_node.py:126-135
    q = vd.mean(0) + torch.randn(2, 2048, 3, generator=g).cuda() * torch.tensor([0.06, 0.02, 0.05]).cuda()
    out = geo.mesh_sdf(q, vd, fd)
    ref = go.compute_mano_cano_sdf(vd.cpu().double()[None].expand(2, -1, -1), fd.cpu(), q.cpu().double())
    away = ref.abs() > 1e-5  # the sign of points on the surface is undefined
    assert float((out.cpu().double() - ref)[away].abs().max()) < 2e-6
    assert 0.2 < float((ref < 0).double().mean()) < 0.8  # the test sees both sides
    off, ins = geo.check_off_in_surface_points_cano_mesh(vd, fd, q, 2 * 256, threshold=0.01)
    roff, rins = go.check_off_in_surface_points_cano_mesh(vd.cpu().double()[None].expand(2, -1, -1), fd.cpu(),
                                                          q.cpu().double(), 2 * 256, 0.01)
    assert torch.equal(off.cpu(), roff) and torch.equal(ins.cpu(), rins)


def test_mesh_index_off_surface_equals_brute_force():
    """the grid-accelerated per-ray test (hold_ray_off_surface + MeshIndex) makes the decisions of evaluating the exact
    signed distance at every sample (hold_mesh_sdf -> min over the ray > threshold), on the loss-target meshes the
    reference uses: the sealed + subdivided MANO at threshold 0.01 and a marching-tetrahedra object mesh at 0.05."""
    from hold_amd import fitting as ft, geometry as geo, meshing as M, synthetic as syn
    g = torch.Generator().manual_seed(3)
    m = syn.make_mano_model(True)
    vs, fs = ft.seal_mano_mesh(torch.tensor(m["v_template"], dtype=torch.float32)[None].cuda(), torch.tensor(m["f"]).cuda(), True)
    vd, fd = geo.subdivide_loop(vs[0], fs)
    r = 0.35
    sph = M.generate_mesh(lambda x: {"sdf": x.norm(dim=1) - r}, np.array([[-r] * 3, [r] * 3]), res_init=32, res_up=1)
    cases = [(vd, fd, 0.01, 0.12), (torch.tensor(sph.vertices, dtype=torch.float32).cuda(), torch.tensor(sph.faces).cuda(), 0.05, 0.6)]
    for verts, faces, thr, reach in cases:
        idx = geo.MeshIndex(verts, faces, thr)
        assert idx.h * 3 ** 0.5 < thr
        n_rays, S = 3000, 98
        ctr = verts.mean(0)
        o = ctr + torch.randn(n_rays, 1, 3, generator=g).cuda() * reach  # ray-like sample sets: a segment through the neighbourhood
        d = torch.nn.functional.normalize(torch.randn(n_rays, 1, 3, generator=g).cuda(), dim=-1)
        t = torch.sort(torch.rand(n_rays, S, 1, generator=g).cuda() * 2 * reach - reach, dim=1).values
        pts = torch.zeros(n_rays * S, 4, device="cuda")
        pts[:, :3] = (o + t * d).reshape(-1, 3)
        fast = idx.off_surface(pts[:, :3], n_rays)  # strided view, as HOLDNet passes its canonical points
        brute, _ = geo.check_off_in_surface_points_cano_mesh(verts, faces, pts[:, :3].reshape(1, -1, 3).contiguous(), n_rays,
                                                             threshold=thr)
        assert 0.05 < float(fast.float().mean()) < 0.95, float(fast.float().mean())
        assert int((fast != brute).sum()) <= 1, (thr, int((fast != brute).sum()))  # at most an fp tie at the threshold


def test_loop_subdivision_matches_oracle():
    from hold_amd import fitting as ft, geometry as geo, synthetic as syn
    from oracle import targets_oracle as to
    m = syn.make_mano_model(False)
    v = torch.tensor(m["v_template"], dtype=torch.float32)[None]
    f = torch.tensor(m["f"])
    vs, fs = ft.seal_mano_mesh(v, f, False)
    vd, fd = geo.subdivide_loop(vs[0].cuda(), fs.cuda())
    ov, of = to.subdivide_loop(vs[0].numpy(), fs.numpy())
    from oracle import meshing_oracle as mo
    ok, why = mo.triangles_match(list(vd.cpu().numpy()[fd.cpu().numpy()]), list(ov[of]), atol=1e-6)  # fp32 vs fp64
    assert ok, why


# ---------------------------------------------------------------------------------------------- f-4 meshing
def test_marching_tetrahedra_kernel_matches_oracle():
    from hold_amd import meshing as M
    from oracle import meshing_oracle as mo
    rs = np.random.RandomState(0)
    n = 11
    ax = np.linspace(-1, 1, n)
    X, Y, Z = np.meshgrid(ax, ax, ax, indexing="ij")
    fields = {"sphere": np.sqrt(X ** 2 + Y ** 2 + Z ** 2) - 0.63, "noise": rs.randn(n, n, n),
              "torus": (np.sqrt(X ** 2 + Y ** 2) - 0.5) ** 2 + Z ** 2 - 0.09}
    for name, vals in fields.items():
        vals = vals.astype(np.float32)
        v, f = M.marching_tetrahedra(torch.from_numpy(vals).cuda(), (-1.0, -1.0, -1.0), 2.0 / (n - 1))
        tris = v.cpu().numpy()[f.cpu().numpy()]
        ok, why = mo.triangles_match(list(tris), mo.marching_tetrahedra(vals.astype(np.float64), [-1, -1, -1], 2.0 / (n - 1)))
        assert ok, (name, why)
        if name != "noise":
            assert mo.mesh_stats(v.cpu().numpy(), f.cpu().numpy())["closed_oriented"], name


def test_generate_mesh_sphere_and_node_meshing(ctx):
    """generate_mesh (meshing.py:9-72) at the shipped resolution on a closed-form sphere, then the two nodes' own
    meshing_cano() on the learnt SDFs (watertight, outward, vertices on the level set)."""
    from hold_amd import meshing as M
    r = 0.31
    mesh = M.generate_mesh(lambda x: {"sdf": x.norm(dim=1) - r}, np.array([[-r, -r, -r], [r, r, r]]), res_init=64, res_up=1)
    v, f = np.asarray(mesh.vertices), np.asarray(mesh.faces)
    from oracle import meshing_oracle as mo
    st = mo.mesh_stats(v, f)
    assert st["closed_oriented"]
    assert abs(st["volume"] / (4 / 3 * math.pi * r ** 3) - 1) < 2e-3 and abs(st["area"] / (4 * math.pi * r ** 2) - 1) < 2e-3
    assert np.abs(np.linalg.norm(v, axis=1) - r).max() < 2e-4
    net = hip_net(ctx["sc"], ctx["sd_np"])
    for nid in ("object", "right"):
        node = net.nodes[nid]
        if nid == "object":
            m = node.meshing_cano()  # object_node.py:112-121: bbox of pts.cano x 2, res 128, updates the loss-target mesh
        else:
            # the synthetic hand SDF is the reference's untrained geometric init (a sphere of radius ~0.5): its level set
            # pokes out of MANO's canonical box (mano_node.py:143), so node.meshing_cano() gives an OPEN patch at best, as
            # the reference would at initialisation; mesh it in a box that contains it for the closed-surface checks
            node.meshing_cano()
            m = M.generate_mesh(lambda x: {"sdf": node.implicit_network.sdf(x)}, np.array([[-1.0] * 3, [1.0] * 3]),
                                res_init=64, res_up=1)
        v, f = np.asarray(m.vertices), np.asarray(m.faces)
        st = mo.mesh_stats(v, f)
        assert f.shape[0] > 1000 and st["closed_oriented"] and st["volume"] > 0, nid
        s = node.implicit_network.sdf(torch.from_numpy(v).float().cuda())
        assert float(s.abs().max()) < 1e-2, nid  # linear interpolation error of the grid (spacing ~0.017)
    obj = net.nodes["object"]
    assert obj.mesh_o is not None and obj.mesh_o.shape[1:] == (obj.mesh_fo_cano.shape[0], 3, 3)


# ---------------------------------------------------------------------------------------------- a13 + f-2 + f-3
def _sphere_mesh(r, n=24):
    from hold_amd import meshing as M
    m = M.generate_mesh(lambda x: {"sdf": x.norm(dim=1) - r}, np.array([[-r, -r, -r], [r, r, r]]), res_init=n, res_up=0)
    return m


def test_training_forward_emits_loss_targets_and_full_loss_matches_oracle(ctx):
    """steady-state training step (step % 200 == 0 -> spawn, object mesh present): HOLDNet.forward emits
    index_off_surface / pts2mano_sdf_cano / pred_sdf / grad_theta (hold_utils.py:149-240), hold_amd.loss.Loss == the
    oracle's Loss restatement term by term, and d loss / d parameters == torch autograd on the oracle."""
    from hold_amd import meshing as M_
    from hold_amd.loss import Loss
    from oracle import targets_oracle as to
    sc, sd, osc = ctx["sc"], ctx["sd"], ctx["osc"]
    sdg = {k: (v.clone().requires_grad_(True) if v.dtype.is_floating_point else v) for k, v in sd.items()}
    W, frames = 6, [0, 2]
    b, oinp = oracle_input(sc, sdg, frames, W, W)
    N = len(frames) * W * W
    rng = _rng(sc, N)
    step, epoch = 400, 25
    net = hip_net(sc, ctx["sd_np"], train=True)
    # loss-target mesh of the object: a small sphere inside its SDF blob, so that central rays come within 0.05 of it
    # and outer rays do not (the blob itself fills the 6x6-ray field of view)
    net.nodes["object"].update_cano(_sphere_mesh(0.08))
    inp = hip_input(b, net, epoch=epoch, step=step)
    out = net(inp, rng=_cuda_rng(rng))
    for k in ("right.index_off_surface", "right.grad_theta", "right.pts2mano_sdf_cano", "right.pred_sdf",
              "object.index_off_surface", "object.grad_theta", "step", "epoch"):
        assert k in out, k
    assert out["right.grad_theta"].shape == (2, 307, 3) and out["right.pred_sdf"].shape == (2, 307)
    assert out["right.index_off_surface"].shape == (N,) and out["right.index_off_surface"].dtype == torch.bool
    assert hasattr(out, "search") and len(out.search("index_off_surface")) == 2
    fracs = {nid: float(out[f"{nid}.index_off_surface"].float().mean()) for nid in ("right", "object")}
    assert all(0.0 < f < 1.0 for f in fracs.values()), fracs  # both kinds of rays for both nodes
    assert net.nodes["right"].mesh_v_cano_div.shape == (3110, 3)
    # ---- oracle on the HIP sampler's z_vals and the HIP-drawn sample points
    zo = {n: out[n + ".z_vals"].detach().cpu() for n in sc["entities"]}
    ex = {}
    oo = ho.holdnet_forward(osc, sdg, oinp, True, rng=rng, z_override=zo, current_epoch=epoch, barf_alpha_iter=4000,
                            extras=ex, stable_merge=True)
    oo["step"], oo["epoch"] = step, epoch
    hand, obj = net.nodes["right"], net.nodes["object"]
    B = len(frames)
    bw = ho.barf_weights(4000, 6, 3)
    tg = to.loss_targets_hand(sdg, "right", hand.mesh_v_cano_div.cpu(), hand.mesh_f_cano_div.cpu(),
                              ex["right"]["x_c"].detach().view(B, -1, 3), hand._last_targets["mano_cano_samples"].cpu(),
                              hand._last_targets["eikonal_samples"].cpu(), N)
    tg.update(to.loss_targets_object(sdg, "object", obj.mesh_vo_cano[0].cpu(), obj.mesh_fo_cano.cpu(),
                                     ex["object"]["x_c"].detach().view(B, -1, 3), obj._last_targets["eikonal_samples"].cpu(),
                                     N, embed_w=bw))
    oo.update(tg)
    assert float((out["right.pts2mano_sdf_cano"].cpu() - tg["right.pts2mano_sdf_cano"]).abs().max()) < 2e-6
    assert rel_err(out["right.pred_sdf"], tg["right.pred_sdf"]) < 1e-4
    for nid in ("right", "object"):
        assert rel_err(out[f"{nid}.grad_theta"], tg[f"{nid}.grad_theta"]) < 1e-4, nid
        assert float((out[f"{nid}.index_off_surface"].cpu() != tg[f"{nid}.index_off_surface"]).float().mean()) < 0.02, nid
        oo[f"{nid}.index_off_surface"] = out[f"{nid}.index_off_surface"].cpu()  # borderline rays: same index set for the loss
    # ---- Loss, term by term
    batch_o = {"gt.rgb": torch.from_numpy(b["gt.rgb"]), "gt.mask": torch.from_numpy(b["gt.mask"])}
    lo = to.loss_forward(batch_o, oo)
    lh = Loss()(inp, out)
    for k in lo:  # a node without a single off-surface ray gives mean(empty) = NaN in the reference (loss_terms.py:44-56) and here
        assert float(lh[k]) == pytest.approx(float(lo[k]), rel=1e-4, abs=1e-7, nan_ok=True), k
    assert all(math.isfinite(float(v)) for v in lo.values()), (fracs, {k: float(v) for k, v in lo.items()})
    lo["loss"].backward()
    lh["loss"].backward()
    checked = 0
    for name, p in net.named_parameters():
        if name not in sdg or sdg[name].grad is None:
            continue
        og = sdg[name].grad
        assert p.grad is not None, name
        rel = float((p.grad.cpu() - og).norm() / (og.norm() + 1e-20))
        assert rel < 1e-3, (name, rel)
        checked += 1
    assert checked >= 100
    # the BARF counter stepped once
    assert int(net.nodes["object"].implicit_network.embedder_obj.alpha_iter) == 4001


def test_second_forward_before_backward_is_detected(ctx):
    """activations live in per-node pools: a backward after a later forward of the same net must raise, not return
    silently wrong gradients (round-1 advisor finding)."""
    sc = ctx["sc"]
    b, _ = oracle_input(sc, ctx["sd"], [0], 4, 4)
    net = hip_net(sc, ctx["sd_np"], train=True)
    out1 = net(hip_input(b, net, epoch=25, step=1))
    net(hip_input(b, net, epoch=25, step=2))
    with pytest.raises(RuntimeError, match="overwritten by a later forward"):
        out1["rgb"].sum().backward()


def test_train_step_steps_barf_once_and_chunks_sum(ctx):
    from hold_amd.loss import Loss
    from hold_amd.train import train_step
    sc = ctx["sc"]
    from hold_amd import synthetic as syn
    net = hip_net(sc, ctx["sd_np"], train=True)
    b = syn.make_batch(sc, [0], syn.make_uv(8, 8), 8, 8)
    inp = {k: torch.from_numpy(v).cuda() for k, v in b.items()}
    it0 = int(net.nodes["object"].implicit_network.embedder_obj.alpha_iter)
    loss, n = train_step(net, inp, 16, step=7, epoch=25, loss_fn=Loss())
    assert n == 64 and math.isfinite(loss)
    assert int(net.nodes["object"].implicit_network.embedder_obj.alpha_iter) == it0 + 1  # 4 chunks, one step


# ---------------------------------------------------------------------------------------------- f-3 optimiser
def test_flat_adam_matches_torch_adam_with_clipping(ctx):
    """FlatAdam (hold_sumsq + hold_adam_step on one bucket) == clip_grad_norm_(0.5) + torch.optim.Adam with the
    reference's two learning-rate groups (hold.py:79-101, train.py:30), over several steps."""
    from hold_amd.optim import FlatAdam, split_params
    sc = ctx["sc"]
    net = hip_net(sc, ctx["sd_np"], train=True)
    ref = hip_net(sc, ctx["sd_np"], train=True)  # same state (weight-normed modules cannot be deep-copied)
    low, main = split_params(ref)
    topt = torch.optim.Adam([{"params": low, "lr": 5e-5}, {"params": main, "lr": 5e-4}], lr=5e-4, eps=1e-8)
    opt = FlatAdam(net, lr=5e-4, clip_norm=0.5)
    pr = dict(ref.named_parameters())
    names = [n for n, p in net.named_parameters() if p.requires_grad and p.numel()]
    pn = dict(net.named_parameters())
    g = torch.Generator(device="cuda").manual_seed(3)
    for it in range(4):
        opt.zero_grad()
        for n in names:
            gr = torch.randn(pn[n].shape, device="cuda", generator=g) * (10.0 if it % 2 == 0 else 1e-5)  # clipped / not clipped (norm 0.015)
            pn[n].grad.copy_(gr)
            pr[n].grad = gr.clone()
        norm_ref = torch.nn.utils.clip_grad_norm_([pr[n] for n in names], 0.5)
        assert opt.grad_norm() == pytest.approx(float(norm_ref), rel=1e-5)
        topt.step()
        opt.step()
        worst = max(float((pn[n].detach() - pr[n].detach()).abs().max()) for n in names)
        assert worst < 2e-7, (it, worst)
    # parameters are views of one bucket, the model still runs and its gradients land in the bucket
    b, _ = oracle_input(sc, ctx["sd"], [0], 4, 4)
    opt.zero_grad()
    out = net(hip_input(b, net, epoch=25, step=1))
    out["rgb"].mean().backward()
    opt.gather_stray_grads()
    assert float(opt.grad.abs().sum()) > 0 and opt.grad_norm() > 0


# ---------------------------------------------------------------------------------------------- (b) per-module surface
def test_reference_call_surface_of_the_modules(ctx):
    """ImplicitNet.forward / gradient, ErrorBoundSampler.get_z_vals(sdf_fn, deformer, implicit_network, ...),
    MANODeformer.forward / forward_skinning, ObjectDeformer, sdf_func_with_deformer -- reference signatures
     against the oracle."""
    from hold_amd import volsdf_utils as VU
    sc, sd, osc = ctx["sc"], ctx["sd"], ctx["osc"]
    net = hip_net(sc, ctx["sd_np"])
    b, oinp = oracle_input(sc, sd, [1, 3], 6, 6)
    inp = hip_input(b, net)
    g = torch.Generator().manual_seed(1)
    # ImplicitNet.forward / gradient
    for nid in ("right", "object"):
        inet = net.nodes[nid].implicit_network
        x = torch.rand(2, 100, 3, generator=g) * 0.4 - 0.2
        cond = None if nid == "object" else torch.zeros(200, 45)
        o = ho.implicit_net(sd, f"nodes.{nid}.implicit_network", x.reshape(-1, 3), cond, 6, None, zero_cond=nid != "object")
        h = inet(x.cuda(), {"pose": torch.zeros(2, 45).cuda()})
        assert h.shape == (2, 100, 257) and rel_err(h.reshape(-1, 257), o) < 1e-4
        gh = inet.gradient(x.reshape(-1, 3).cuda(), None)
        xo = x.reshape(-1, 3).clone().requires_grad_(True)
        so = ho.implicit_net(sd, f"nodes.{nid}.implicit_network", xo, cond, 6, None, zero_cond=nid != "object")[:, :1]
        go_ = torch.autograd.grad(so, xo, torch.ones_like(so))[0]
        assert gh.shape == (200, 1, 3) and rel_err(gh[:, 0], go_) < 1e-4
    # deformers
    hand = net.nodes["right"]
    so = hand.server(inp["right.params"][:, 0], inp["right.transl"], inp["right.full_pose"], inp["right.betas"])
    pts = so["verts"][:, :300] + torch.randn(2, 300, 3, generator=g).cuda() * 0.01
    xc, outlier = hand.deformer.forward(pts, so["tfs"], return_weights=False, inverse=True, verts=so["verts"])
    full_pose = torch.cat([oinp["right.global_orient"], oinp["right.pose"]], 1)
    oso = ho.mano_server(osc.mano["right"], osc.tfs_c_inv["right"], oinp["right.params"][:, 0], oinp["right.transl"],
                         full_pose, oinp["right.betas"])
    w, dmin = ho.query_skinning_weights(pts.cpu(), oso["verts"], osc.skin_w["right"])
    oxc = ho.skinning(pts.cpu(), w, oso["tfs"], inverse=True)
    assert rel_err(xc, oxc) < 1e-5 and outlier.shape == (2, 300) and not bool(outlier.any())
    back = hand.deformer.forward_skinning(xc, None, so["tfs"])
    wc, _ = ho.query_skinning_weights(oxc, osc.verts_c["right"].expand(2, -1, -1), osc.skin_w["right"])
    assert rel_err(back, ho.skinning(oxc, wc, oso["tfs"], inverse=False)) < 1e-5
    obj = net.nodes["object"]
    T = obj.server.object_model(rot=inp["object.global_orient"], trans=inp["object.transl"],
                                scene_scale=inp["object.params"][:, 0], want_verts=False)["T"]
    q = torch.randn(2, 50, 3, generator=g).cuda()
    xo_, _ = obj.deformer.forward(q, T, inverse=True)
    assert rel_err(obj.deformer.forward_skinning(xo_, None, T), q) < 1e-5
    # sdf_func_with_deformer + the sampler's reference entry point
    deform_info = {"cond": {"pose": torch.zeros(2, 45).cuda()}, "tfs": so["tfs"], "verts": so["verts"]}
    sdf, x_c, feat = VU.sdf_func_with_deformer(hand.deformer, hand.implicit_network, False, pts.reshape(-1, 3), deform_info)
    assert sdf.shape == (2, 300, 1) and x_c.shape == (2, 300, 3) and feat.shape == (2, 300, 256)
    ray_dirs, cam_loc = net_rays(inp)
    z_ref = hand.ray_sampler.get_z_vals(VU.sdf_func_with_deformer, hand.deformer, hand.implicit_network, ray_dirs,
                                        cam_loc, hand.density, False, deform_info)
    out = net(inp)
    assert torch.equal(z_ref, out["right.z_vals"])  # same kernels behind both entry points
    z_gen = hand.ray_sampler.get_z_vals(lambda d, n, t, x, di: VU.sdf_func_with_deformer(d, n, t, x, di), hand.deformer,
                                        hand.implicit_network, ray_dirs, cam_loc, hand.density, False, deform_info)
    assert float(((z_gen - z_ref).abs() > 1e-3).float().mean()) < 0.02  # generic-callable route (layer-wise trunk)


def net_rays(inp):
    from hold_amd import kernels as K
    return K.raygen(inp["uv"], inp["extrinsics"], inp["intrinsics"])


def test_record_hip_training_output_fixture(ctx, tmp_path):
    """scripts/record_hip_outputs.py runs end to end (its output, copied to tests/golden/hip_train_output.npz, is what
    tests/test_dropin_cpu.py feeds to the reference's own Loss)."""
    import subprocess
    import sys
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = tmp_path / "hip_train_output.npz"
    subprocess.run([sys.executable, os.path.join(root, "scripts", "record_hip_outputs.py"), str(out)], check=True)
    z = np.load(out, allow_pickle=True)
    assert "out.right.index_off_surface" in z.files and "loss.loss" in z.files


def test_chunked_loss_terms_add_up_to_the_unchunked_loss():
    """ray-chunked steps (hold_amd.train.train_step): every ray-wise term of the full Loss -- rgb, semantics and the
    opacity-sparsity mean over the OFF-SURFACE rays of the whole batch (loss_terms.py:44-56) -- must add up over the chunks
    to the un-chunked loss, with the same gradients, in steady state (the sparsity term is normalised by the previous
    step's whole-batch count); a chunk without a single off-surface ray contributes zero, not 0 / 0 (round-2 advisor)."""
    from hold_amd.loss import Loss
    dev = "cuda:0"
    N, C = 2048, 4
    g = torch.Generator().manual_seed(11)
    r = lambda *s: torch.rand(*s, generator=g)
    leaves = {"rgb": r(N, 3), "semantics": r(N, 4), "right.mask_prob": r(N), "object.mask_prob": r(N)}
    off = {"right": r(N) > 0.5, "object": r(N) > 0.7}
    off["object"][N // C:2 * N // C] = False  # the second chunk has no off-surface ray of this node
    batch = {"gt.rgb": r(N, 3).to(dev), "gt.mask": torch.tensor([0, 50, 150, 250])[torch.randint(0, 4, (N,), generator=g)].to(dev)}

    def run(chunks, loss_fn):
        lv = {k: v.clone().to(dev).requires_grad_(True) for k, v in leaves.items()}
        tot = {}
        for c in range(chunks):
            sl = slice(c * N // chunks, (c + 1) * N // chunks)
            out = {k: v[sl] for k, v in lv.items()}
            out.update({f"{n}.index_off_surface": o[sl].to(dev) for n, o in off.items()}, step=15000)
            b = {k: v[sl] for k, v in batch.items()}
            if chunks > 1:
                b["hold_amd.n_total"], b["hold_amd.frame_terms"] = N, c == 0
            ld = loss_fn(b, out)
            ld["loss"].backward()
            for k, v in ld.items():
                tot[k] = tot.get(k, 0.0) + float(v.detach() if torch.is_tensor(v) else v)
        return tot, {k: v.grad.clone() for k, v in lv.items()}

    full, g_full = run(1, Loss())
    chunked = Loss()
    run(C, chunked)                   # first step: populates the whole-batch counts
    part, g_part = run(C, chunked)    # steady state
    for k in ("loss/rgb", "loss/sem", "loss/opacity_sparse", "loss"):
        assert math.isfinite(part[k]) and part[k] == pytest.approx(full[k], rel=2e-5), (k, part[k], full[k])
    for k in g_full:
        assert torch.isfinite(g_part[k]).all(), k
        assert float((g_part[k] - g_full[k]).abs().max()) <= 2e-5 * float(g_full[k].abs().max()), k


def test_ray_tile_losses_of_two_ranks_add_up_to_the_unsharded_loss():
    """bench.py --split rays / train_step(n_total = frame rays): two ranks own the two ray tiles of ONE frame and their
    gradients are SUMMED (FlatAdam averages, grad_mul = world undoes it).  For the FULL Loss -- ray-wise terms, the
    opacity-sparsity mean over the frame's off-surface rays, and the per-frame terms (eikonal, MANO-canonical SDF) that
    every rank evaluates -- the two ranks' losses and gradients must add up to the un-sharded step's, from the first step
    on for the per-frame terms and in steady state for the sparsity term (round-3 advisor: rank-local previous counts made
    it world x too large from step 2 on, the per-frame terms were counted once per rank)."""
    from hold_amd.loss import Loss
    dev = "cuda:0"
    N, W = 2048, 2
    g = torch.Generator().manual_seed(12)
    r = lambda *s: torch.rand(*s, generator=g)
    leaves = {"rgb": r(N, 3), "semantics": r(N, 4), "right.mask_prob": r(N), "object.mask_prob": r(N)}
    # per-frame terms: the same tensors on every rank (replicated weights, same eikonal / canonical samples)
    # (gradient norms around 12: the eikonal term passes the reference's `> 0.0008` gate, loss.py:86-88)
    frame_leaves = {"right.grad_theta": r(1, 307, 3) * 14, "object.grad_theta": r(1, 307, 3) * 14, "right.pred_sdf": r(1, 307) * 0.02 - 0.01}
    frame_const = {"right.pts2mano_sdf_cano": (r(1, 307) * 0.02 - 0.01).to(dev)}
    off = {"right": r(N) > 0.5, "object": r(N) > 0.8}  # uneven over the tiles: rank-local counts differ from the frame's
    off["object"][:N // 2] &= r(N // 2) > 0.5
    batch = {"gt.rgb": r(N, 3).to(dev), "gt.mask": torch.tensor([0, 50, 150, 250])[torch.randint(0, 4, (N,), generator=g)].to(dev)}

    def run(rank_slices, loss_fns):
        lv = {k: v.clone().to(dev).requires_grad_(True) for k, v in {**leaves, **frame_leaves}.items()}
        tot = {}
        for sl, loss_fn in zip(rank_slices, loss_fns):
            out = {k: lv[k][sl] for k in leaves}
            out.update({k: lv[k] for k in frame_leaves})
            out.update(frame_const)
            out.update({f"{n}.index_off_surface": o[sl].to(dev) for n, o in off.items()}, step=15000)
            b = {k: v[sl] for k, v in batch.items()}
            if len(rank_slices) > 1:  # what train_step puts into the batch of a rank that owns one ray tile (one chunk)
                b["hold_amd.n_total"], b["hold_amd.frame_terms"], b["hold_amd.rays_owned"] = N, True, sl.stop - sl.start
            ld = loss_fn(b, out)
            ld["loss"].backward()
            for k, v in ld.items():
                tot[k] = tot.get(k, 0.0) + float(v.detach() if torch.is_tensor(v) else v)
        return tot, {k: v.grad.clone() for k, v in lv.items()}

    full, g_full = run([slice(0, N)], [Loss()])
    assert full["loss/eikonal"] > 0 and full["loss/mano_cano"] > 0 and full["loss/opacity_sparse"] > 0
    tiles = [slice(0, N // W), slice(N // W, N)]
    ranks = [Loss(), Loss()]
    first, g_first = run(tiles, ranks)   # first step: the sparsity denominator is each tile's count scaled to the frame
    for k in ("loss/rgb", "loss/sem", "loss/eikonal", "loss/mano_cano"):
        assert first[k] == pytest.approx(full[k], rel=2e-5), (k, first[k], full[k])
    part, g_part = run(tiles, ranks)     # steady state
    for k in ("loss/rgb", "loss/sem", "loss/eikonal", "loss/mano_cano", "loss"):
        assert part[k] == pytest.approx(full[k], rel=2e-5 if k != "loss" else 2e-2), (k, part[k], full[k])
    # the sparsity term: every rank estimates the frame's count from its own tile (prev * world): the estimate differs from
    # the true count by the tiles' imbalance, the SUM over ranks stays within it (it was `world` x off before)
    cnt = {n: float(o.sum()) for n, o in off.items()}
    imb = max(abs(float(o[t].sum()) * W - cnt[n]) / cnt[n] for n, o in off.items() for t in tiles)
    assert part["loss/opacity_sparse"] == pytest.approx(full["loss/opacity_sparse"], rel=imb / (1 - imb) + 1e-4), (part, full, imb)
    for k in g_full:
        tol = (imb / (1 - imb) + 1e-3) if "mask_prob" in k else 2e-5  # 1 / (estimated count) against 1 / (true count)
        assert float((g_part[k] - g_full[k]).abs().max()) <= tol * float(g_full[k].abs().max()), k
    # with the counts summed over the ranks (Loss.sync_group: one scalar all-reduce per node and step; here the hook it goes
    # through returns the two tiles' sum) the steady-state step is EXACT
    ranks = [Loss(), Loss()]
    run(tiles, ranks)
    totals = {n: torch.tensor(float(o.sum()), device=dev) for n, o in off.items()}
    for lf in ranks:
        acc = dict(lf._off_acc)
        lf.count_reduce = (lambda t, _acc=acc: next(totals[n] for n, v in _acc.items() if v is t))
    part, g_part = run(tiles, ranks)
    for k in ("loss/rgb", "loss/sem", "loss/eikonal", "loss/mano_cano", "loss/opacity_sparse", "loss"):
        assert part[k] == pytest.approx(full[k], rel=2e-5), (k, part[k], full[k])
    for k in g_full:
        assert float((g_part[k] - g_full[k]).abs().max()) <= 2e-5 * float(g_full[k].abs().max()), k


def test_five_step_trajectory_matches_oracle_with_torch_adam(ctx):
    """every other training parity test is ONE step; this one carries the cross-step state --
    the BARF counter (4000 .. 4004), the spawned canonical MANO (forced at the first step, step % 200 == 0), the per-node
    activation pools and their generation counters, FlatAdam's moments and step count, the weight-pack invalidation after a
    HIP-side parameter update, the speculative sampler's round prediction -- through FIVE optimiser steps: HOLDNet.forward
    (HIP sampler in the loop, speculation on) + hold_amd.loss.Loss + backward + FlatAdam.step(clip 0.5) against five oracle
    steps (oracle forward on the HIP path's z_vals and sample points, the oracle's Loss restatement, torch autograd,
    clip_grad_norm_(0.5), torch.optim.Adam with the reference's parameter groups: code/src/hold/hold.py:79-101,
    code/train.py:30), fresh random draws every step.  Held: the loss curve (1e-5 relative per step), the sampler's round
    counts against the ORACLE's own sampler at the oracle's weights of that step (equal in >= 4 of 5 steps, see below), and after step 5 every parameter tensor
    within 1e-4 of its norm (measured 4.5e-5) -- and, the sharper statement, every tensor's five-step UPDATE within 5e-3 of the update's norm
    (Adam divides by sqrt(v): elements whose gradient is below eps = 1e-8 move by amounts that depend on their last bits)."""
    from hold_amd.loss import Loss
    from hold_amd.optim import FlatAdam
    from oracle import targets_oracle as to
    sc, sd, osc = ctx["sc"], ctx["sd"], ctx["osc"]
    sdg = {k: (v.clone().requires_grad_(True) if v.dtype.is_floating_point else v.clone()) for k, v in sd.items()}
    W, frames, epoch = 6, [0, 2], 25
    N = len(frames) * W * W
    net = hip_net(sc, ctx["sd_np"], train=True)
    net.nodes["object"].update_cano(_sphere_mesh(0.08))
    opt = FlatAdam(net, lr=5e-4, clip_norm=0.5)
    names = [n for n, p in net.named_parameters() if p.requires_grad and p.numel() and n in sdg]
    assert len(names) >= 100
    node_names = [n for n in names if ".params." in n]
    topt = torch.optim.Adam([{"params": [sdg[n] for n in node_names], "lr": 5e-5},
                             {"params": [sdg[n] for n in names if n not in node_names], "lr": 5e-4}], lr=5e-4, eps=1e-8)
    p0 = {n: sdg[n].detach().clone() for n in names}
    hand, obj = net.nodes["right"], net.nodes["object"]
    B = len(frames)
    batch_o = None
    losses, round_log = [], []
    for k in range(5):
        step = 400 + k  # k = 0: step % 200 == 0 -> spawn_cano_mano at the first step
        rng = _rng(sc, N, seed=11 + k)
        b, oinp = oracle_input(sc, sdg, frames, W, W)
        batch_o = {"gt.rgb": torch.from_numpy(b["gt.rgb"]), "gt.mask": torch.from_numpy(b["gt.mask"])}
        # ---- HIP step
        inp = hip_input(b, net, epoch=epoch, step=step)
        opt.zero_grad()
        out = net(inp, rng=_cuda_rng(rng))
        assert int(obj.implicit_network.embedder_obj.alpha_iter) == 4001 + k  # the BARF counter steps once per training forward
        rounds = {n: net.nodes[n].ray_sampler.last_iters for n in sc["entities"]}
        lh = Loss()(inp, out)
        lh["loss"].backward()
        # ---- the oracle's own sampler at the oracle's weights of this step: the number of rounds.  A round's convergence test is
        # a threshold on the WORST ray's error bound (ray_sampler.py:244): two trajectories that differ by fp32 rounding can
        # take it differently on a borderline ray, so the counts are collected here and held below to "equal in at least four
        # of the five steps, never more than one apart" (the rest of the step uses the HIP path's z_vals on both sides)
        ex_s = {}  # (detached copies: no graph; the oracle's normal path needs autograd enabled for d sdf / d x)
        ho.holdnet_forward(osc, {kk: (v.detach() if torch.is_tensor(v) else v) for kk, v in sdg.items()},
                           {kk: (v.detach() if torch.is_tensor(v) else v) for kk, v in oinp.items()}, True, rng=rng,
                           current_epoch=epoch, barf_alpha_iter=4000 + k, extras=ex_s)
        round_log.append({n: (ex_s[n]["iters"], rounds[n]) for n in sc["entities"]})
        # ---- oracle step on the HIP path's z_vals and drawn sample points
        zo = {n: out[n + ".z_vals"].detach().cpu() for n in sc["entities"]}
        ex = {}
        oo = ho.holdnet_forward(osc, sdg, oinp, True, rng=rng, z_override=zo, current_epoch=epoch, barf_alpha_iter=4000 + k,
                                extras=ex, stable_merge=True)
        oo["step"], oo["epoch"] = step, epoch
        bw = ho.barf_weights(4000 + k, 6, 3)
        tg = to.loss_targets_hand(sdg, "right", hand.mesh_v_cano_div.cpu(), hand.mesh_f_cano_div.cpu(),
                                  ex["right"]["x_c"].detach().view(B, -1, 3), hand._last_targets["mano_cano_samples"].cpu(),
                                  hand._last_targets["eikonal_samples"].cpu(), N)
        tg.update(to.loss_targets_object(sdg, "object", obj.mesh_vo_cano[0].cpu(), obj.mesh_fo_cano.cpu(),
                                         ex["object"]["x_c"].detach().view(B, -1, 3), obj._last_targets["eikonal_samples"].cpu(),
                                         N, embed_w=bw))
        oo.update(tg)
        for nid in ("right", "object"):  # borderline rays of the off-surface test: the same index set for both losses
            oo[f"{nid}.index_off_surface"] = out[f"{nid}.index_off_surface"].cpu()
        lo = to.loss_forward(batch_o, oo)
        losses.append((float(lo["loss"]), float(lh["loss"])))
        assert float(lh["loss"]) == pytest.approx(float(lo["loss"]), rel=1e-5), (k, losses)
        topt.zero_grad()
        lo["loss"].backward()
        torch.nn.utils.clip_grad_norm_([sdg[n] for n in names], 0.5)
        topt.step()
        opt.step()
    assert opt.step_count == 5
    for n in sc["entities"]:
        pairs = [r[n] for r in round_log]
        assert all(abs(a - b) <= 1 for a, b in pairs) and sum(a == b for a, b in pairs) >= 4, (n, pairs)
    worst_p, worst_d = ("", 0.0), ("", 0.0)
    pn = dict(net.named_parameters())
    for n in names:
        ph, po = pn[n].detach().cpu().double(), sdg[n].detach().double()
        rp = float((ph - po).norm() / (po.norm() + 1e-30))
        dh, do = ph - p0[n].double(), po - p0[n].double()
        rd = float((dh - do).norm() / (do.norm() + 1e-30)) if float(do.norm()) > 0 else 0.0
        worst_p = max(worst_p, (n, rp), key=lambda t: t[1])
        worst_d = max(worst_d, (n, rd), key=lambda t: t[1])
    print(f"five-step trajectory: losses {losses}; sampler rounds (oracle, HIP) per step {round_log}; worst parameter error {worst_p}; "
          f"worst update error {worst_d}")
    assert worst_p[1] < 1e-4, worst_p
    assert worst_d[1] < 5e-3, worst_d  # measured 1.3e-3 (round 5, GPU call 9)
