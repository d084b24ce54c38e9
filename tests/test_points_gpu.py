"""GPU: the K = 15 nearest-vertex selection / skinning-weight blend of hold_knn_invlbs_fwd (threshold -> filter ->
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
