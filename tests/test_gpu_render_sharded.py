"""The N > 1 render path end to end with two processes: each builds its own scene replica and renderer (here both on
GPU 0 -- the box has one), renders its own sample range through shard.render_sharded, and the colour buffers are summed
with a gloo all-reduce (RCCL needs one GPU per rank; bench.py's render leg uses it on a node).  The sum equals a single
renderer's 2 S samples up to the order of the float additions."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

SPP = 3


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _make():
    from craytracer_amd import api, scenes
    si, desc = scenes.cornell29_render(max_depth=4, seed=8)
    cam, flm = scenes.cornell_camera(80, 60)
    built = api.BuiltScene(si, api.ACCEL_NONE)
    scene = api.Scene(built, 0)
    return api.Renderer(scene, desc, cam, flm), scene, built


def _worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    from craytracer_amd import shard
    dist.init_process_group("gloo", rank=rank, world_size=world)
    r, scene, built = _make()
    buf, total = shard.render_sharded(r, SPP, rank, world)
    np.savez(os.path.join(out_dir, "r%d.npz" % rank), buf=buf.numpy(), total=total)
    r.close()
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_render_one_image(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    got = [np.load(os.path.join(str(tmp_path), "r%d.npz" % k)) for k in range(world)]
    assert int(got[0]["total"]) == int(got[1]["total"]) == world * SPP
    assert np.array_equal(got[0]["buf"], got[1]["buf"])
    r, scene, built = _make()
    r.render(world * SPP)
    whole, n = r.read()
    # and the two halves, rendered by this process, add up to exactly what the ranks reduced
    r.reset(0); r.render(SPP); lo, _ = r.read()
    r.reset(SPP); r.render(SPP); hi, _ = r.read()
    r.close()
    assert n == world * SPP
    assert np.array_equal(got[0]["buf"].reshape(whole.shape), lo + hi)
    assert np.allclose(got[0]["buf"].reshape(whole.shape), whole, rtol=1e-5, atol=1e-6)


# ---- tile sharding: bands of the textured scene, all-gathered ----
TILE_W, TILE_H, TILE_SPP = 72, 50, 3


def _make_tiled(rows):
    from craytracer_amd import api, scenes
    si, desc, pos, look = scenes.textured_scene(rows=rows)
    cam = api.make_camera(pos, look, focal_pt_dist=0.05)
    flm = api.make_film(cam, 1.1, TILE_W, TILE_H)
    built = api.BuiltScene(si, api.ACCEL_BVH4)
    scene = api.Scene(built, 0)
    return api.Renderer(scene, desc, cam, flm), scene, built


def _tile_worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    from craytracer_amd import shard
    dist.init_process_group("gloo", rank=rank, world_size=world)
    r, scene, built = _make_tiled(shard.tile_rows(TILE_H, rank, world))
    frame = shard.render_tiled(r, TILE_SPP, TILE_H, TILE_W, rank, world)
    np.save(os.path.join(out_dir, "t%d.npy" % rank), frame.numpy())
    r.close()
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_render_two_bands_of_the_textured_scene(tmp_path):
    """BASELINE.json configs[3] stand-in, tile-sharded: two processes (each its own scene replica and renderer, both on the
    box's one GPU) render the upper and the lower band, all-gather them over gloo, and both hold the whole frame -- bit for
    bit the frame one renderer renders alone."""
    world = 2
    mp.spawn(_tile_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    got = [np.load(os.path.join(str(tmp_path), "t%d.npy" % k)) for k in range(world)]
    assert got[0].shape == (TILE_H, TILE_W, 3) and np.array_equal(got[0], got[1])
    r, scene, built = _make_tiled(None)
    r.render(TILE_SPP)
    whole, n = r.read()
    r.close()
    assert n == TILE_SPP
    assert np.array_equal(got[0].view(np.uint32), whole.view(np.uint32))
