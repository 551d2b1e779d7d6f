"""CPU tests of the multi-GPU path (batch sharding + result gather) with world_size 2 over gloo."""
import os
import socket
import sys

import pytest

from cdc_compression_amd.parallel import shard_bounds, sharded_decode

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_bounds_cover_batch_exactly():
    for B in (1, 2, 7, 32, 33, 256):
        for world in (1, 2, 3, 4, 8):
            spans = [shard_bounds(B, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == B
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_bounds(4, 2, 2)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, B, q):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(0)                         # every rank holds the same full batch
    init = torch.randn(B, 3, 8, 8)
    ctx = [torch.randn(B, 4, 8, 8), torch.randn(B, 6, 4, 4)]

    def fake_decode(i, c):                       # stands in for cdc_decode: a per-image function
        return i * 2.0 + c[0][:, :3] - c[1].mean(dim=(1, 2, 3), keepdim=True)

    out = sharded_decode(fake_decode, init, ctx, world, rank, dist)
    ref = fake_decode(init, ctx)
    q.put((rank, bool(torch.equal(out, ref)), tuple(out.shape)))
    dist.destroy_process_group()


@pytest.mark.parametrize("B", [4, 5])
def test_sharded_decode_gathers_full_batch_world2(B):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, B, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(r[0] for r in res) == [0, 1]
    assert all(ok for _, ok, _ in res)
    assert all(shape == (B, 3, 8, 8) for _, _, shape in res)
