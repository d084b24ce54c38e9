"""hold_weight_norm_fwd / _bwd (csrc/wnorm.hip) through hold_net._effective_all against the per-layer autograd graph of
torch.nn.utils.weight_norm's formula w = v * (g / ||v||_row) (code/src/networks/shape_net.py:79-80, texture_net.py:40-41)."""
import pytest
import torch
import torch.nn as nn

from parity_common import ROOT  # noqa: F401

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _lins():
    torch.manual_seed(0)
    lins = [nn.utils.weight_norm(nn.Linear(k, n)) for n, k in [(256, 84), (256, 256), (217, 256), (257, 256), (3, 256)]]
    lins.append(nn.Linear(8, 5))  # a plain layer passes through
    lins = [l.to(DEV) for l in lins]
    with torch.no_grad():
        for l in lins[:5]:
            l.weight_g.mul_(torch.rand_like(l.weight_g) + 0.5)
    return lins


def test_fused_weight_norm_matches_per_layer_autograd():
    from hold_amd import hold_net as H
    lins = _lins()
    ref, new = [H._eff(l) for l in lins], H._effective_all(lins)
    for a, b in zip(ref, new):  # the row norm is summed in another order: rounding-level differences only
        assert a.shape == b.shape and float((a - b).abs().max()) <= 2e-6 * float(a.abs().max())
        assert b.data_ptr() % 16 == 0 and b.is_contiguous()
    cot = [torch.randn_like(r) for r in ref]
    ps = [p for l in lins for p in l.parameters()]
    g1 = torch.autograd.grad(sum((a * c).sum() for a, c in zip(ref, cot)), ps, allow_unused=True)
    g3 = torch.autograd.grad(sum((a * c).sum() for a, c in zip(H._effective_all(lins), cot)), ps, allow_unused=True)
    for a, b in zip(g1, g3):
        assert (a is None) == (b is None)
        if a is not None:
            assert float((a - b).abs().max()) <= 1e-5 * float(a.abs().max() + 1e-12)
    # an unused output: that layer's parameters receive no gradient (None), the others are unaffected
    new = H._effective_all(lins)
    g2 = torch.autograd.grad(sum((a * c).sum() for a, c in zip(new[:4], cot[:4])), ps, allow_unused=True)
    unused = [i for i, p_ in enumerate(ps) if p_ is lins[4].weight_v][0]
    assert g2[unused] is None
    used = [i for i, p_ in enumerate(ps) if p_ is lins[1].weight_v][0]
    assert torch.equal(g2[used], g3[used])


def test_gradients_of_bucket_parameters_are_added_in_place():
    """parameters marked as living in FlatAdam's gradient bucket: the backward kernel adds into p.grad itself (twice here:
    the sum of two backward passes), and what autograd would have accumulated is identical"""
    from hold_amd import hold_net as H
    lins = _lins()
    cot = [torch.randn(l.weight_v.shape if hasattr(l, "weight_v") else l.weight.shape, device=DEV) for l in lins]
    loss = lambda: sum((a * c).sum() for a, c in zip(H._effective_all(lins), cot))
    ps = [p for l in lins[:5] for p in (l.weight_v, l.weight_g)]
    loss().backward()
    loss().backward()
    want = [p.grad.clone() for p in ps]
    for p in ps:
        p.grad = torch.full_like(p, 0.5)  # a pre-existing value must be kept (accumulation, not a store)
        p._hold_bucket = True
    loss().backward()
    loss().backward()
    for p, w in zip(ps, want):
        assert float((p.grad - 0.5 - w).abs().max()) <= 1e-5 * float(w.abs().max() + 1e-12)
