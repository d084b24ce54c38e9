"""world_size-2 gloo test of the data-parallel path: allreduce(grad of shard_i) == grad of the full batch."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from parity_common import ROOT  # noqa: F401  (sys.path)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _model():
    torch.manual_seed(0)
    m = torch.nn.Sequential(torch.nn.Linear(6, 16), torch.nn.Softplus(beta=100), torch.nn.Linear(16, 3))
    table = torch.nn.Embedding(4, 6)  # per-frame pose table: each rank touches only its own rows
    return m, table


def _worker(rank, world, port, out):
    from hold_amd import parallel

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    m, table = _model()
    frames = parallel.shard_frames(list(range(4)), rank, world)
    x = table(torch.tensor(frames))
    loss = m(x).pow(2).sum() / 4  # the global batch has 4 frames
    loss.backward()
    params = list(m.parameters()) + list(table.parameters())
    nbytes = parallel.allreduce_grads(params, average=False)
    assert nbytes == sum(p.numel() for p in params) * 4
    if rank == 0:
        torch.save([p.grad.clone() for p in params], out)
    dist.destroy_process_group()


def test_allreduce_of_shards_equals_full_batch_grad(tmp_path):
    out = str(tmp_path / "g.pt")
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    got = torch.load(out)
    m, table = _model()
    loss = m(table(torch.arange(4))).pow(2).sum() / 4
    loss.backward()
    ref = [p.grad for p in list(m.parameters()) + list(table.parameters())]
    for a, b in zip(got, ref):
        assert torch.allclose(a, b, atol=1e-6), (a - b).abs().max()


def test_shard_frames_partition():
    from hold_amd import parallel

    ids = list(range(10))
    parts = [parallel.shard_frames(ids, r, 4) for r in range(4)]
    assert sum(parts, []) == ids


def _holdnet_worker(rank, world, port, out):
    """the real model's parameter set through the flat bucket: every trainable parameter of HOLDNet (dense nets, density
    betas, frame latents, per-frame pose tables) gets a rank-dependent synthetic gradient; ONE all-reduce of the bucket
    must leave every parameter's .grad (a view of the bucket) equal to the sum over ranks."""
    import hold_amd
    from hold_amd import synthetic as syn
    from hold_amd.optim import FlatAdam

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(0)
    sc = syn.make_scene(n_frames=4)
    net = hold_amd.build_from_scene(sc, syn.make_state_dict(sc), device="cpu")
    for node in net.nodes.values():
        node.params.defrost()
    opt = FlatAdam(net, lr=5e-4, clip_norm=0.5)
    assert opt.n_params > 2_190_000  # dense nets + density betas + latents + pose tables
    low = sum(p.numel() for node in net.nodes.values() for p in node.params.parameters())
    assert low == 4 * (3 + 3 + 45) + 10 + 4 * 6 and opt.n_low >= low
    base = opt.grad.data_ptr()
    for p, off in zip(opt.params, opt.offsets):  # .grad and .data are 256-byte aligned views of the two buckets
        assert off % 64 == 0 and p.grad.data_ptr() == base + 4 * off and p.data.data_ptr() == opt.flat.data_ptr() + 4 * off
    opt.zero_grad()
    g = torch.Generator().manual_seed(7)
    vals = [torch.randn(p.shape, generator=g) for p in opt.params]  # same draws on every rank
    for p, v in zip(opt.params, vals):
        p.grad.add_(v * (rank + 1))  # autograd-style in-place accumulation into the view
    # a pose-table row only this rank touched (frames are sharded): other ranks contribute zeros
    tr = net.nodes["right"].params.transl.weight
    tr.grad[rank] += 100.0 * (rank + 1)
    scale = opt.allreduce(average=True)
    assert scale == 1.0 / world
    if rank == 0:
        torch.save(dict(grads=[p.grad.clone() for p in opt.params], vals=vals,
                        row=tr.grad[:2].clone(), base=tr.grad.data_ptr() - opt.grad.data_ptr()), out)
    dist.destroy_process_group()


def test_flat_bucket_allreduce_over_holdnet_parameters(tmp_path):
    out = str(tmp_path / "h.pt")
    mp.spawn(_holdnet_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    d = torch.load(out)
    for gsum, v in zip(d["grads"][5:], d["vals"][5:]):  # (the first entries hold the specially marked pose rows)
        assert torch.allclose(gsum, v * 3.0, atol=1e-5)
    assert d["base"] >= 0  # the pose table's gradient lives inside the bucket


# ------------------------------------------------------------------------------------------ ray-tile sharding (strong scaling)
def _rays(n):
    """a fan of rays from (0, 0, -2): the first half hits a 0.5-sphere at the origin, the second half misses it"""
    g = torch.Generator().manual_seed(4)
    t = torch.cat([torch.rand(n // 2, generator=g) * 0.15, 0.45 + torch.rand(n - n // 2, generator=g) * 0.3])
    ph = torch.rand(n, generator=g) * 6.28318
    d = torch.stack([torch.sin(t) * torch.cos(ph), torch.sin(t) * torch.sin(ph), torch.cos(t)], -1)
    return d, torch.tensor([0.0, 0.0, -2.0]).expand(n, 3).contiguous()


def _sample(dirs, cam, sync=None):
    from oracle import hold_oracle as ho
    R = 3.0
    far = ho.sphere_far(cam, dirs, R)
    z0 = ho.uniform_z(torch.zeros(dirs.shape[0], 1), far, 128, None)
    sdf = lambda p: p.norm(dim=-1) - 0.5
    return ho.error_bound_sample(z0, sdf, cam, dirs, torch.tensor(0.0101), R, False, None, sync=sync)


def _tile_worker(rank, world, port, out):
    from hold_amd.sampler import ErrorBoundSampler

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dirs, cam = _rays(64)
    lo, hi = rank * 32, (rank + 1) * 32
    smp = ErrorBoundSampler(3.0)
    smp.sync_group = True
    # the host-side exchange of the HIP sampler itself: (max beta, error flag) -> MAX over the ranks
    mb, err = smp.sync_round(0.25 * (rank + 1), rank == 1)
    assert mb == 0.5 and err is True
    assert smp.sync_round(0.125, False) == (0.125, False)
    z_sync, it_sync = _sample(dirs[lo:hi], cam[lo:hi], sync=lambda b: smp.sync_round(b, False)[0])
    z_solo, it_solo = _sample(dirs[lo:hi], cam[lo:hi])
    torch.save(dict(z_sync=z_sync, it_sync=it_sync, z_solo=z_solo, it_solo=it_solo), out + str(rank))
    dist.destroy_process_group()


def test_ray_tiles_with_synchronised_rounds_equal_the_unsharded_call(tmp_path):
    """SURVEY 8(e) caveat: the sampler's convergence test is a max over ALL rays of a call.  Two ranks that own ray tiles of
    one frame and exchange that max (ErrorBoundSampler.sync_round: one 2-float MAX all-reduce per round) produce exactly the
    z_vals and round count of the un-sharded call; without the exchange the tile whose rays converge early stops early."""
    out = str(tmp_path / "t")
    mp.spawn(_tile_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    dirs, cam = _rays(64)
    z_full, it_full = _sample(dirs, cam)
    parts = [torch.load(out + str(r)) for r in range(2)]
    assert all(p["it_sync"] == it_full for p in parts)
    assert torch.equal(torch.cat([p["z_sync"] for p in parts]), z_full)
    assert min(p["it_solo"] for p in parts) < it_full  # the test distinguishes the two behaviours


# ------------------------------------------------------------------------------------------ world 4 and 8 (first contact of an 8-GPU run)
def test_ray_tiles_partition_every_frame_size():
    from hold_amd.parallel import ray_tile
    for n in (8, 67, 1000, 512 * 512, 512 * 512 + 5):
        for world in (1, 2, 3, 4, 8):
            tiles = [ray_tile(n, r, world) for r in range(world)]
            assert tiles[0][0] == 0 and tiles[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(tiles, tiles[1:]))
            sizes = [hi - lo for lo, hi in tiles]
            assert max(sizes) - min(sizes) <= 1 and min(sizes) >= 1


class _ToyNode(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.params = torch.nn.Embedding(1, 3)  # one frame's pose row: EVERY tile's rays depend on it


class _ToyNet(torch.nn.Module):
    """what FlatAdam needs of HOLDNet: .nodes[*].params (pose tables, 0.1 lr) + dense parameters"""

    def __init__(self):
        super().__init__()
        torch.manual_seed(3)
        self.nodes = torch.nn.ModuleDict({"right": _ToyNode()})
        self.mlp = torch.nn.Sequential(torch.nn.Linear(3, 16), torch.nn.Softplus(beta=100), torch.nn.Linear(16, 3))

    def loss_sum(self, rays, gt):
        """ray-wise L1 term, SUM over the given rays (the caller divides by the frame's ray count)"""
        x = rays + self.nodes["right"].params.weight[0]
        return (self.mlp(x) - gt).abs().sum()


def _frame(n):
    g = torch.Generator().manual_seed(9)
    return torch.randn(n, 3, generator=g), torch.rand(n, 3, generator=g)


def _split_rays_worker(rank, world, port, n, out):
    """one optimiser-step's gradient exchange of bench.py --split rays: rank r owns ray_tile(n, r, world) of ONE frame, the
    loss is normalised by the frame's ray count, FlatAdam.allreduce averages over the ranks and grad_mul = world undoes
    it -- the bucket must then hold the un-sharded frame gradient on every rank, whatever the tile sizes"""
    from hold_amd.optim import FlatAdam
    from hold_amd.parallel import ray_tile

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    net = _ToyNet()
    opt = FlatAdam(net, lr=5e-4, clip_norm=0.5)
    opt.zero_grad()
    rays, gt = _frame(n)
    lo, hi = ray_tile(n, rank, world)
    (net.loss_sum(rays[lo:hi], gt[lo:hi]) / n).backward()
    opt.gather_stray_grads()
    grad_mul = float(world) * opt.allreduce(average=True)
    assert dist.get_world_size() == world
    torch.save(dict(grad=opt.grad * grad_mul, tile=(lo, hi)), out + str(rank))
    dist.destroy_process_group()


def _check_split_rays(tmp_path, world, n):
    from hold_amd.optim import FlatAdam

    out = str(tmp_path / f"s{world}_")
    mp.spawn(_split_rays_worker, args=(world, _free_port(), n, out), nprocs=world, join=True)
    net = _ToyNet()
    opt = FlatAdam(net, lr=5e-4, clip_norm=0.5)
    opt.zero_grad()
    rays, gt = _frame(n)
    (net.loss_sum(rays, gt) / n).backward()
    opt.gather_stray_grads()
    parts = [torch.load(out + str(r)) for r in range(world)]
    assert parts[0]["tile"][0] == 0 and parts[-1]["tile"][1] == n
    for p in parts:  # every rank ends with the same, complete gradient (incl. the shared pose row)
        assert torch.allclose(p["grad"], opt.grad, atol=2e-6, rtol=1e-5), float((p["grad"] - opt.grad).abs().max())
    assert float(opt.grad[:3].abs().sum()) > 0  # the pose row is inside the bucket and received gradient from every tile


def test_split_rays_world4_non_divisible_tiles(tmp_path):
    _check_split_rays(tmp_path, 4, 1003)  # tiles of 250 / 251 rays


def test_split_rays_world8_last_tiles_short(tmp_path):
    _check_split_rays(tmp_path, 8, 67)   # 8 or 9 rays per rank


def _tile_worker_n(rank, world, port, n, out):
    from hold_amd.parallel import ray_tile
    from hold_amd.sampler import ErrorBoundSampler

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dirs, cam = _rays(n)
    lo, hi = ray_tile(n, rank, world)
    smp = ErrorBoundSampler(3.0)
    smp.sync_group = True
    z, it = _sample(dirs[lo:hi], cam[lo:hi], sync=lambda b: smp.sync_round(b, False)[0])
    torch.save(dict(z=z, it=it), out + str(rank))
    dist.destroy_process_group()


def test_ray_tiles_world4_uneven_tiles_run_the_unsharded_rounds(tmp_path):
    """the sampler's per-round MAX exchange with 4 ranks and tiles of unequal size (70 rays: 17 / 18): every tile runs the
    round count of the un-sharded call and the concatenated z_vals are bit-identical to it"""
    out = str(tmp_path / "u")
    mp.spawn(_tile_worker_n, args=(4, _free_port(), 70, out), nprocs=4, join=True)
    dirs, cam = _rays(70)
    z_full, it_full = _sample(dirs, cam)
    parts = [torch.load(out + str(r)) for r in range(4)]
    assert all(p["it"] == it_full for p in parts)
    assert torch.equal(torch.cat([p["z"] for p in parts]), z_full)


# ------------------------------------------------------------------------------------------ same number of forwards on every rank
def test_tile_chunks_give_every_rank_the_same_number_of_forwards():
    """every forward of a ray tile is a sequence of collectives (sampler rounds, loss counts); with
    tiles that differ by one ray, ceil(tile / chunk) of each rank's OWN tile can differ across ranks when a tile size
    straddles a multiple of the chunk (world 3 / 5 / 7, tile ~ k chunk).  parallel.tile_chunks + chunk_bounds: the same
    count on every rank, balanced non-empty chunks that fit chunk_rays and cover the tile."""
    from hold_amd.parallel import chunk_bounds, ray_tile, tile_chunks
    hit = 0
    for world in (1, 2, 3, 4, 5, 7, 8):
        for chunk in (16, 100, 16384):
            for k in (1, 2, 3):
                for d in (-3, -1, 0, 1, 2, world - 1, world, world + 1):
                    n = world * k * chunk + d
                    if n < world:
                        continue
                    tiles = [hi - lo for lo, hi in (ray_tile(n, r, world) for r in range(world))]
                    own = {-(-t // chunk) for t in tiles}
                    hit += len(own) > 1  # the situation that deadlocked: ranks disagree on ceil(own tile / chunk)
                    nc = tile_chunks(n, world, chunk)
                    assert nc == max(own)
                    for t in tiles:
                        b = chunk_bounds(t, nc)
                        assert len(b) == nc and b[0][0] == 0 and b[-1][1] == t
                        assert all(b[i][1] == b[i + 1][0] for i in range(nc - 1))
                        assert all(0 < hi - lo <= chunk for lo, hi in b)
    assert hit > 20  # the cases above do contain the disagreement
    assert tile_chunks(98305, 3, 16384) == 3 and [-(-(hi - lo) // 16384) for lo, hi in (ray_tile(98305, r, 3) for r in range(3))] == [2, 2, 3]
    # a tile with fewer rays than forwards would leave a rank an empty chunk: refused from numbers EVERY rank shares, i.e. by
    # all ranks together and before any collective (advisor r5: one rank raising alone is a hang for the others)
    with pytest.raises(ValueError):
        tile_chunks(5, 4, 1)  # tiles 1, 1, 1, 2 -> 2 forwards, three ranks own one ray


def _chunked_tile_worker(rank, world, port, n, chunk, out):
    from hold_amd.parallel import chunk_bounds, ray_tile, tile_chunks
    from hold_amd.sampler import ErrorBoundSampler

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dirs, cam = _rays(n)
    lo, hi = ray_tile(n, rank, world)
    smp = ErrorBoundSampler(3.0)
    smp.sync_group = True
    calls, rounds = 0, []
    for a, b in chunk_bounds(hi - lo, tile_chunks(n, world, chunk)):  # one sampler call (a run of per-round collectives) per chunk
        z, it = _sample(dirs[lo + a:lo + b], cam[lo + a:lo + b], sync=lambda m: smp.sync_round(m, False)[0])
        calls += 1
        rounds.append(it)
    t = torch.tensor([float(calls)])
    dist.all_reduce(t)  # the step's gradient all-reduce: reached by every rank after the SAME number of sampler calls
    torch.save(dict(calls=calls, rounds=rounds, total=float(t)), out + str(rank))
    dist.destroy_process_group()


def test_ray_tiles_world3_tiles_straddling_a_chunk_multiple_do_not_deadlock(tmp_path):
    """3 ranks, 97 rays in chunks of 16: tiles of 32 / 32 / 33 rays -- ceil(own tile / 16) = 2, 2, 3.  With the common chunk
    count every rank makes 3 sampler calls (each a run of per-round MAX exchanges), agrees on the rounds of every call, and
    reaches the final all-reduce."""
    out = str(tmp_path / "c")
    mp.spawn(_chunked_tile_worker, args=(3, _free_port(), 97, 16, out), nprocs=3, join=True)
    parts = [torch.load(out + str(r)) for r in range(3)]
    assert [p["calls"] for p in parts] == [3, 3, 3] and all(p["total"] == 9.0 for p in parts)
    assert parts[0]["rounds"] == parts[1]["rounds"] == parts[2]["rounds"]
