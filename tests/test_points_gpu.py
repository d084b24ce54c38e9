"""GPU: per-point kernels.  The K = 15 nearest-vertex selection / skinning-weight blend of hold_knn_invlbs_fwd (threshold -> filter ->
selection passes) against a brute-force fp64 top-15 of the same op (code/src/model/mano/deformer.py:145-170,
pytorch3d knn_points K = 15)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n_verts,shared", [(778, False), (778, True), (61, False), (800, False)])
def test_knn_blend_matches_bruteforce_top15(n_verts, shared):
    from hold_amd import kernels as K
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(n_verts)
    B, ppf = 3, 5000
    nf = 1 if shared else B
    verts = (torch.randn(nf, n_verts, 3, generator=g) * 0.08).to(dev)
    skin = torch.softmax(torch.randn(n_verts, 16, generator=g) * 3, 1).to(dev).contiguous()
    # points: half near the vertex cloud, half far away (all distances nearly equal: the widest filter sets)
    P = B * ppf
    x = torch.zeros(P, 4, device=dev)
    x[:, :3] = (torch.randn(P, 3, generator=g) * 0.1).to(dev)
    x[P // 2:, :3] += torch.tensor([1.5, -0.7, 2.0], device=dev)
    w = torch.empty(P, 16, device=dev)
    K.knn_invlbs(x, P, ppf, verts if not shared else verts[0], skin, w_out=w)
    torch.cuda.synchronize()
    vd = verts.double()
    v_pp = vd[torch.arange(P, device=dev) // ppf] if not shared else vd.expand(P, -1, -1)
    d2 = ((x[:, None, :3].double() - v_pp) ** 2).sum(-1)  # [P, V]
    ds, idx = torch.sort(d2, dim=1, stable=True)
    conf = torch.exp(-ds[:, :15].clamp(max=4.0))
    conf = conf / conf.sum(1, keepdim=True)
    ref = (conf[:, :, None] * skin.double()[idx[:, :15]]).sum(1)
    err = (w.double() - ref).abs().max(1).values
    tie = (ds[:, 15] - ds[:, 14]) <= 1e-6 * ds[:, 14]  # fp32 cannot order these: either choice is the reference's
    assert int(tie.sum()) < 5
    assert float(err[~tie].max()) < 1e-5, (float(err[~tie].max()), int((err > 1e-5).sum()))
    assert float((w.sum(1) - 1).abs().max()) < 1e-5


def _embed64(x, L, bw):
    """the positional embedding [x, sin(2^k x), cos(2^k x)]_k with per-column weights (embedder.py:31-49, BARF weights
    hold_utils / embedder 'barf' mode), fp64 torch, differentiable"""
    cols = [x]
    for k in range(L):
        cols += [torch.sin(x * 2.0 ** k), torch.cos(x * 2.0 ** k)]
    return torch.cat(cols, 1) * bw


@pytest.mark.parametrize("barf", [False, True])
def test_embed_backward_and_double_backward_match_autograd(barf):
    """hold_embed_bwd (g = (dE/dx)^T ge) and hold_embed_bwd2 (its double backward: gebar and the second-derivative term of
    xbar) against fp64 autograd -- with `ge` living in a 256-wide buffer's columns 217.. and the second copy of gebar written
    IN PLACE over it (what field.py does for the ascending sweep's side columns)"""
    from hold_amd import kernels as K
    dev = torch.device("cuda:0")
    torch.manual_seed(3 + barf)
    P, L = 4099, 6
    E = 3 + 6 * L
    x = torch.zeros(P, 4, device=dev)
    x[:, :3] = torch.randn(P, 3, device=dev) * 0.7
    bw = (torch.rand(E, device=dev) if barf else None)
    host = torch.randn(P, 256, device=dev)          # stands for t_3: ge = its columns 217..
    ge = host[:, 217:]
    assert ge.shape[1] == E
    ge0, host0 = ge.clone(), host.clone()
    gbar = torch.zeros(P, 4, device=dev)
    gbar[:, :3] = torch.randn(P, 3, device=dev)
    # fp64 reference
    x64 = x[:, :3].double().requires_grad_(True)
    ge64 = ge0.double().requires_grad_(True)
    bw64 = bw.double() if barf else torch.ones(E, device=dev, dtype=torch.float64)
    (g64,) = torch.autograd.grad((_embed64(x64, L, bw64) * ge64).sum(), x64, create_graph=True)
    gebar64, xbar64 = torch.autograd.grad((g64 * gbar[:, :3].double()).sum(), (ge64, x64))
    # first order
    g = torch.zeros(P, 4, device=dev)
    K.embed_bwd(x, L, P, ge, g, barf_w=bw)
    assert float((g[:, :3].double() - g64.detach()).abs().max()) < 2e-4 * float(g64.abs().max())
    # second order, second copy in place over ge
    gebar = torch.full((P, 40), 7.0, device=dev)
    xbar = torch.ones(P, 4, device=dev)
    K.embed_bwd2(x, L, P, ge, gbar, gebar, xbar=xbar, barf_w=bw, gebar2=ge)
    torch.cuda.synchronize()
    sc = float(gebar64.abs().max())
    assert float((gebar[:, :E].double() - gebar64).abs().max()) < 2e-5 * sc
    assert torch.all(gebar[:, E:] == 7.0)
    assert torch.equal(host[:, 217:], gebar[:, :E])                 # the in-place copy is the same bits
    assert torch.equal(host[:, :217], host0[:, :217]) and not torch.equal(ge, ge0)
    assert float((xbar[:, :3].double() - 1 - xbar64).abs().max()) < 2e-5 * float(xbar64.abs().max())
    # and equals the run with separate buffers bit for bit
    gebar_b = torch.empty(P, 40, device=dev)
    xbar_b = torch.ones(P, 4, device=dev)
    K.embed_bwd2(x, L, P, ge0, gbar, gebar_b, xbar=xbar_b, barf_w=bw)
    assert torch.equal(gebar_b[:, :E], gebar[:, :E]) and torch.equal(xbar_b, xbar)


def test_seed_dsp_vector_and_scalar_paths():
    """hold_seed_dsp: t = w * softplus'(h) recovered from h = softplus(a) (beta = 100): the 16-byte path (N % 4 == 0, aligned
    rows) and the element path (a weight vector at a 4-byte-aligned address) give the same bits, both equal fp64 to 1e-6"""
    from hold_amd import kernels as K
    dev = torch.device("cuda:0")
    torch.manual_seed(11)
    P = 10007
    a = torch.randn(P, 256, device=dev) * 0.05
    h = torch.nn.functional.softplus(a, beta=100)
    w_al = torch.randn(256, device=dev)
    w_un = torch.empty(260, device=dev)[1:257]  # 4-byte aligned only: the element path
    w_un.copy_(w_al)
    t1 = torch.full((P, 256), 7.0, device=dev)
    t2 = torch.full((P, 256), 7.0, device=dev)
    K.seed_dsp(h, w_al, 256, P, t1)
    K.seed_dsp(h, w_un, 256, P, t2)
    torch.cuda.synchronize()
    assert torch.equal(t1, t2)
    ref = w_al.double() * torch.sigmoid(100 * a.double())
    assert float((t1.double() - ref).abs().max()) < 1e-5 * float(ref.abs().max())
