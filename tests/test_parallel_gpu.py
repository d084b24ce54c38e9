"""Two ranks on ONE GPU (gloo over device tensors): FlatAdam.step -- all-reduce of the flat gradient bucket, clip on the
REDUCED gradients, Adam (hold_sumsq + hold_adam_step) -- must equal the single-rank step on the averaged gradients, and the
ranks must end bit-identical (replicated weights stay replicated).  The 8-GPU RCCL run is the driver's; this covers the
ordering and the device-side plumbing of the N > 1 path with real kernels."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from parity_common import ROOT  # noqa: F401

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _net_opt(group_ok=True):
    import hold_amd
    from hold_amd import synthetic as syn
    from hold_amd.optim import FlatAdam
    torch.manual_seed(0)
    sc = syn.make_scene(n_frames=4)
    net = hold_amd.build_from_scene(sc, syn.make_state_dict(sc), device="cuda:0")
    for node in net.nodes.values():
        node.params.defrost()
    return net, FlatAdam(net, lr=5e-4, clip_norm=0.5)


def _grads(opt, rank):
    g = torch.Generator().manual_seed(100 + rank)
    return [torch.randn(p.shape, generator=g) * (0.3 + rank) for p in opt.params]


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    net, opt = _net_opt()
    for _ in range(2):  # two steps: the second one exercises the moments
        opt.zero_grad()
        for p, g in zip(opt.params, _grads(opt, rank)):
            p.grad.add_(g.cuda())
        opt.step()
    torch.cuda.synchronize()
    torch.save(opt.flat.cpu(), out + str(rank))
    dist.destroy_process_group()


def test_two_rank_flat_adam_step_equals_single_rank_on_averaged_gradients(tmp_path):
    out = str(tmp_path / "p")
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    p0, p1 = torch.load(out + "0"), torch.load(out + "1")
    assert torch.equal(p0, p1)
    net, opt = _net_opt()
    for _ in range(2):
        opt.zero_grad()
        for p, g0, g1 in zip(opt.params, _grads(opt, 0), _grads(opt, 1)):
            p.grad.add_(((g0 + g1) * 0.5).cuda())
        opt.step()
    ref = opt.flat.cpu()
    assert (p0 - ref).abs().max().item() <= 1e-7 + 1e-6 * ref.abs().max().item()
