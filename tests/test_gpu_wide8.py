"""The 8-wide walk (crt_wide8.h, an experiment switch: CRT_WIDE8=1 at scene creation) against the oracle: the same hit records
bit for bit, including the rays it hands to the 4-wide walk of the same launch -- exact-t ties, rays whose slab arithmetic can
produce NaN -- and through every kind of launch (plain, counted, slice, with objects outside the tree)."""
import ctypes as C

import numpy as np
import pytest

from craytracer_amd import api, scenes
import oraclelib as ol

pytestmark = pytest.mark.gpu


def _hits(t):
    return t.cpu().numpy().view(np.uint8).reshape(-1, 20).copy().view(api.HIT_DTYPE).reshape(-1)


def _awkward_rays(si, n, seed):
    lo, hi = scenes.scene_bounds(si)
    ext = hi - lo
    rng = np.random.Generator(np.random.PCG64(seed))
    rays = scenes.random_rays(n, lo, hi, seed)
    k = n // 8
    rays[0:k, 0:3] = (lo - ext) + rng.random((k, 3), dtype=np.float32) * (3 * ext)
    rays[k:2 * k] = scenes.axis_parallel_rays(k, lo, hi)                       # zero / -0.0 components: never `clean`
    rays[2 * k:3 * k, 0:3] = hi + np.float32(0.5) * ext
    rays[2 * k:3 * k, 3:6] = np.abs(rays[2 * k:3 * k, 3:6])
    cell = ext / np.float32(16)
    rays[3 * k:4 * k, 0:3] = lo + np.floor(rng.random((k, 3), dtype=np.float32) * 16).astype(np.float32) * cell
    return rays


@pytest.mark.parametrize("sah", [False, True])
@pytest.mark.parametrize("scene", ["blobs", "ties", "soup"])
def test_eight_wide_walk_writes_the_reference_order_records(scene, sah, monkeypatch):
    import torch
    monkeypatch.setenv("CRT_WIDE8", "1")
    if scene == "blobs":
        si = scenes.courtyard(6, 24, 48, seed=13)                  # props touching the floor: a few genuine exact-t ties
    elif scene == "soup":
        si = scenes.random_tris(20000, seed=14)
    else:                                                          # every triangle three times over: EVERY hit is a three-way tie
        v = scenes.random_tris(1500, seed=12, size=0.08).tri_verts
        v = np.concatenate([v, v, v])
        si = api.SceneInput(v[np.random.Generator(np.random.PCG64(4)).permutation(len(v))])
    built = api.BuiltScene(si, api.ACCEL_BVH4, sah=sah)
    sc = api.Scene(built, 0)
    o = ol.OracleScene(si, built=built)
    n = (1 << 18) + 4321                                           # binned, ragged
    rays = _awkward_rays(si, n, 77)
    want = o.intersect(rays, mode=api.TRAVERSE_EXACT)
    assert (want["obj"] >= 0).sum() > n // 10
    sc.set_wide_nodes(True)
    sc.enable_counters(True)
    sc.reset_counters()
    got = sc.intersect(rays)
    c8 = sc.counters()
    assert ol.same_bits(got, want)
    assert c8["rays_closest"] == n                                 # every ray settled exactly once, by one walk or the other
    sc.set_wide_nodes(False)
    sc.reset_counters()
    got4 = sc.intersect(rays)
    c4 = sc.counters()
    assert ol.same_bits(got4, want)
    if scene != "ties":
        assert c8["nodes_visited"] < 0.85 * c4["nodes_visited"]    # fewer, wider visits
    sc.enable_counters(False)
    # device entry points: counted (n on the device, below n_max), slices that tile the batch, a small unbinned batch
    sc.set_wide_nodes(True)
    d_rays = torch.from_numpy(rays).cuda()
    d_n = torch.tensor([n - 1000], dtype=torch.int32, device="cuda")
    out = torch.full((n, 5), -7, dtype=torch.int32, device="cuda")
    sc.intersect_dev_counted(d_rays.data_ptr(), d_n.data_ptr(), n, out.data_ptr(), None)
    torch.cuda.synchronize()
    g = _hits(out)
    assert ol.same_bits(g[:n - 1000], want[:n - 1000])
    assert np.all(out[n - 1000:].cpu().numpy() == -7)              # rays beyond the count are not touched
    out.fill_(-7)
    for k in range(3):
        sc.intersect_dev_slice(d_rays.data_ptr(), n, out.data_ptr(), k, 3, None)
    torch.cuda.synchronize()
    assert ol.same_bits(_hits(out), want)
    out.fill_(-7)
    sc.intersect_dev(d_rays.data_ptr(), 5000, out.data_ptr(), None)
    torch.cuda.synchronize()
    assert ol.same_bits(_hits(out)[:5000], want[:5000])
    sc.close()


def test_eight_wide_copy_is_made_on_request_only():
    si = scenes.random_tris(500, seed=3)
    built = api.BuiltScene(si, api.ACCEL_BVH4)
    sc = api.Scene(built, 0)
    sc.set_wide_nodes(False)
    with pytest.raises(api.CrtError):
        sc.set_wide_nodes(True)
    sc.close()
