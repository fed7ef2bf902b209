"""N>1 path on CPU: world_size-2 gloo job exercising the batch sharding, the max-over-ranks timing reduction
and the output gather that bench.py / the tiled drivers use on RCCL."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp

from dagl_amd.shard import forward_sharded, rank_seed, reduce_max_seconds, shard_range


def test_shard_range_partitions_exactly():
    for n in (0, 1, 2, 7, 8, 64, 65):
        for world in (1, 2, 3, 8):
            spans = [shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            for (a, b), (c, d) in zip(spans, spans[1:]):
                assert b == c
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_range(4, 2, 2)
    assert rank_seed(3, 0) != rank_seed(3, 1)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, B, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.manual_seed(0)
        x = torch.randn(B, 3, 5, 5)                       # identical on every rank
        mod = torch.nn.Conv2d(3, 2, 3, padding=1)         # stand-in for the per-image block (samples independent)
        with torch.no_grad():
            full = mod(x)
            got = forward_sharded(mod, x, dist, gather=True)
        t = reduce_max_seconds(1.0 + rank, dist)
        q.put((rank, bool(torch.allclose(got, full, atol=1e-6, rtol=1e-6)), t))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("B", [2, 5])
def test_world2_gloo_sharded_forward(B):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, B, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, same, t in res:
        assert same, f"rank {rank}: gathered output differs from the unsharded forward"
        assert t == 2.0                                   # max over ranks of (1.0, 2.0)
