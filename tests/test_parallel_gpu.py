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


def _rccl_worker(rank, world, port, out):
    """the collectives of a data-parallel step on the RCCL backend itself (backend "nccl" IS RCCL on ROCm), one rank: the flat
    gradient all-reduce inside FlatAdam.step, the sampler's per-round MAX exchange on a device tensor, a barrier"""
    from hold_amd.sampler import ErrorBoundSampler
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=rank, world_size=world)
    assert dist.get_backend() == "nccl" and dist.get_world_size() == 1
    net, opt = _net_opt()
    before = opt.flat.clone()
    opt.zero_grad()
    for p, g in zip(opt.params, _grads(opt, 0)):
        p.grad.add_(g.cuda())
    gsum = float(opt.grad.double().sum())
    scale = opt.allreduce(average=True)  # one RCCL all-reduce of the bucket: identity with one rank
    assert scale == 1.0 and abs(float(opt.grad.double().sum()) - gsum) <= 1e-9 * abs(gsum)
    opt.step()
    smp = ErrorBoundSampler(3.0)
    smp.sync_group = True
    mb, err = smp.sync_round(0.375, False, dev="cuda:0")
    dist.barrier()
    torch.cuda.synchronize()
    torch.save(dict(moved=float((opt.flat - before).abs().max()), mb=mb, err=err), out)
    dist.destroy_process_group()


def test_rccl_backend_single_rank_step_and_sampler_exchange(tmp_path):
    """the RCCL code path is exercised by the driver's GPU suite every round (the 8-GPU run itself is the
    driver's): process group "nccl" with one rank on the box's GPU -- FlatAdam.step through its all-reduce, the sampler's
    2-float MAX exchange on the device, barrier, teardown."""
    out = str(tmp_path / "rccl")
    mp.spawn(_rccl_worker, args=(1, _free_port(), out), nprocs=1, join=True)
    r = torch.load(out)
    assert r["moved"] > 0 and r["mb"] == 0.375 and r["err"] is False


def test_launch_stream_follows_torchs_current_stream():
    """every entry point is launched on `_lib.stream_ptr()`: it must be torch's CURRENT stream of the current device -- also inside a
    `torch.cuda.stream(side)` block (the overlap of copies / collectives with compute relies on it) -- whichever getter is used."""
    from hold_amd import _lib, kernels as K
    assert (_lib.stream_ptr().value or 0) == torch.cuda.current_stream().cuda_stream
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        assert (_lib.stream_ptr().value or 0) == side.cuda_stream
        # and a kernel launched there is ordered on that stream: it sees the value written just before it on the same stream
        x = torch.full((4096, 4), 3.0, device="cuda")
        y = torch.zeros(4096, 8, device="cuda")
        K.copy_cols(x, y, 4, 4096)
        ev = torch.cuda.Event(); ev.record()
    ev.synchronize()
    assert torch.equal(y[:, :4], x) and float(y[:, 4:].abs().max()) == 0.0
    assert (_lib.stream_ptr().value or 0) == torch.cuda.current_stream().cuda_stream
