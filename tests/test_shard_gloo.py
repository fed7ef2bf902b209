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


def _chop_worker(rank, world, port, shape, ensemble, q):
    import torch.distributed as dist
    from dagl_amd.net import chop_forward, chop_forward_sharded, chop_leaf_boxes
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.manual_seed(3)
        x = torch.randn(*shape)                           # the same full image on every rank
        mod = torch.nn.Sequential(torch.nn.Conv2d(shape[1], 4, 3, padding=1), torch.nn.PReLU(),
                                  torch.nn.Conv2d(4, shape[1], 3, padding=1))     # a cheap seeded conv as the network
        calls = []
        fn = lambda t: (calls.append(t.shape[0]), mod(t))[1]
        with torch.no_grad():
            want = chop_forward(mod, x, ensemble=ensemble)                        # the reference tiling, leaf by leaf
            got = chop_forward_sharded(fn, x, dist, max_batch=1, ensemble=ensemble)    # leaf by leaf, as the reference calls the net
            ran = sum(calls) // (8 if ensemble else 1)
            got5 = chop_forward_sharded(mod, x, dist, max_batch=5, ensemble=ensemble)  # batches of leaves: oneDNN may pick another
                                                                                       # algorithm per batch size -> last-bit differences
        n_leaves = len(chop_leaf_boxes(shape[2], shape[3]))
        q.put((rank, bool(torch.equal(got, want)) and bool(torch.allclose(got5, want, atol=1e-6, rtol=1e-6)), ran, n_leaves))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,shape,ensemble", [(2, (1, 2, 256, 256), False), (3, (2, 1, 200, 184), False), (2, (1, 1, 136, 136), True)])
def test_gloo_sharded_tiler_equals_forward_chop_bit_for_bit(world, shape, ensemble):
    """Tile-level multi-GPU inference (DN_Gray/model/__init__.py:181,195-214): leaves dealt to the ranks, one
    all_gather, every rank stitches -- the same bits as the sequential reference tiling on a cheap conv, and every rank ran only
    its share of the leaves."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_chop_worker, args=(r, world, port, shape, ensemble, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    n_leaves = res[0][3]
    assert sum(r[2] for r in res) == n_leaves * shape[0]                         # the leaves were dealt out, nobody ran them all
    for rank, same, ran, _ in res:
        assert same, f"rank {rank}: stitched output differs from chop_forward"
        assert ran <= (-(-n_leaves // world)) * shape[0]


def test_sharded_tiler_without_a_process_group_is_the_batched_driver():
    from dagl_amd.net import chop_forward, chop_forward_batched, chop_forward_sharded
    torch.manual_seed(1)
    x = torch.randn(1, 1, 144, 160)
    mod = torch.nn.Conv2d(1, 1, 3, padding=1)
    with torch.no_grad():
        got = chop_forward_sharded(mod, x)
        assert torch.equal(got, chop_forward_batched(mod, x))
        assert torch.allclose(got, chop_forward(mod, x), atol=1e-6, rtol=1e-6)
