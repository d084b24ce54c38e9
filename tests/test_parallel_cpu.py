"""world_size-2 gloo test of the data-parallel path: allreduce(grad of shard_i) == grad of the full batch."""
import os
import socket

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
