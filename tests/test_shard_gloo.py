"""N > 1 path on CPU: world_size-2 gloo run of the sharding helpers bench.py uses."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from craytracer_amd import shard


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    b, e = shard.shard_range(n, rank, world)
    # each rank "traces" its slice: a stand-in per-ray value that depends only on the global ray index
    vals = np.arange(b, e, dtype=np.int64) * 3 + 1
    elapsed = 0.010 * (rank + 1)
    worst = shard.max_over_ranks(elapsed)
    total = shard.sum_over_ranks(float(e - b))
    buf = torch.full((4, 5, 3), float(rank + 1), dtype=torch.float32)
    buf, spp = shard.combine_radiance(buf, 8 * (rank + 1))
    # tile sharding: each rank "renders" its band of a 7-row frame (a value that depends only on the pixel), all-gather
    H, W = 7, 5
    rb, re = shard.tile_rows(H, rank, world)
    frame = torch.arange(H * W * 3, dtype=torch.float32).reshape(H, W, 3)
    tiles = shard.gather_tiles(frame[rb:re].clone(), H, W, rank, world)
    np.savez(os.path.join(out_dir, "r%d.npz" % rank), b=b, e=e, vals=vals, worst=worst, total=total,
             buf=buf.numpy(), spp=spp, tiles=tiles.numpy(), rb=rb, re=re)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo(tmp_path):
    n, world = 1001, 2
    mp.spawn(_worker, args=(world, _free_port(), n, str(tmp_path)), nprocs=world, join=True)
    parts = [np.load(tmp_path / ("r%d.npz" % r)) for r in range(world)]
    assert int(parts[0]["b"]) == 0 and int(parts[-1]["e"]) == n
    assert int(parts[0]["e"]) == int(parts[1]["b"])                       # contiguous, no overlap
    allv = np.concatenate([p["vals"] for p in parts])
    assert np.array_equal(allv, np.arange(n, dtype=np.int64) * 3 + 1)     # every ray exactly once
    for p in parts:
        assert abs(float(p["worst"]) - 0.020) < 1e-12                     # max over ranks
        assert float(p["total"]) == n
        assert np.all(p["buf"] == 3.0) and int(p["spp"]) == 24            # 1 + 2, 8 + 16
        assert np.array_equal(p["tiles"], np.arange(7 * 5 * 3, dtype=np.float32).reshape(7, 5, 3))   # every rank holds the whole frame
    assert (int(parts[0]["rb"]), int(parts[0]["re"]), int(parts[1]["rb"]), int(parts[1]["re"])) == (0, 4, 4, 7)   # uneven bands


def test_shard_range_properties():
    for n in (0, 1, 7, 64, 1000003):
        for world in (1, 2, 3, 8):
            edges = [shard.shard_range(n, r, world) for r in range(world)]
            assert edges[0][0] == 0 and edges[-1][1] == n
            sizes = [e - b for b, e in edges]
            assert max(sizes) - min(sizes) <= 1
            for (b0, e0), (b1, e1) in zip(edges, edges[1:]):
                assert e0 == b1
