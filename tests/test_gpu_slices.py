"""crt_intersect_batch_dev_slice / crt_shadow_batch_dev_slice: one batch traced as K shares (what K GPUs do under strong scaling,
SURVEY 8e; the reference's unit of parallel work is a pixel range of a sample pass, main.cpp:276-317).

The shares must TILE the batch -- every ray traced by exactly one share, nothing written outside it -- and the records must be
those of the undivided launch, bit for bit.  A binned batch is cut at cell boundaries (the order inside a cell differs from run to
run, the cell counts do not: each share here comes from its own binning pass, as it would on its own GPU); a batch below the binning
threshold, or a scene without an accelerator, is cut by ray index.  Also here: the ADVICE r2 case of a triangle-only description
with triangles ahead of num_non_grid_obj (the tail-splitting walk must not be chosen for it)."""
import ctypes as C

import numpy as np
import pytest

from craytracer_amd import api, scenes
import oraclelib as ol

pytestmark = pytest.mark.gpu

SENT = 0x7F7F7F7F


def _hits(t):
    return t.cpu().numpy().view(api.HIT_DTYPE).reshape(-1)


def _check_closest(torch, sc, d_rays, n, K, want):
    """K shares, each into its own sentinel-filled buffer: written sets disjoint, union = everything, records = want's."""
    owner = np.full(n, -1, np.int64)
    sizes = []
    for k in range(K):
        got = torch.full((n, 5), SENT, dtype=torch.int32, device="cuda")
        sc.intersect_dev_slice(d_rays.data_ptr(), n, got.data_ptr(), k, K, None)
        torch.cuda.synchronize()
        g = _hits(got)
        written = g["obj"] != SENT
        assert np.all(owner[written] == -1), "a ray was traced by two shares"
        owner[written] = k
        assert ol.same_bits(g[written], want[written])
        # an untouched record is untouched in every word
        assert np.all(got.cpu().numpy()[~written] == SENT)
        sizes.append(int(written.sum()))
    assert np.all(owner >= 0), "a ray was traced by no share"
    return sizes, owner


@pytest.mark.parametrize("accel", [api.ACCEL_BVH4, api.ACCEL_BVH, api.ACCEL_GRID])
@pytest.mark.parametrize("K", [3, 8])
def test_shares_of_a_binned_batch_tile_it(accel, K):
    import torch
    si = scenes.courtyard(16, 32, 64, seed=7)
    built = api.BuiltScene(si, accel, sah=(accel == api.ACCEL_BVH4))
    sc = api.Scene(built, 0)
    lo, hi = scenes.scene_bounds(si)
    n = (1 << 19) + 321
    rays = scenes.random_rays(n, lo, hi, seed=21)
    d_rays = torch.from_numpy(rays).cuda()
    full = torch.empty((n, 5), dtype=torch.int32, device="cuda")
    sc.intersect_dev(d_rays.data_ptr(), n, full.data_ptr(), None)
    torch.cuda.synchronize()
    want = _hits(full)
    o = ol.OracleScene(si, built=built)
    m = 1 << 14
    assert ol.same_bits(want[:m], o.intersect(rays[:m], mode=api.TRAVERSE_PRUNED))
    o.close()
    sizes, owner = _check_closest(torch, sc, d_rays, n, K, want)
    # whole cells of about n / K rays each: no share is empty, none is more than twice its due
    assert min(sizes) > 0 and max(sizes) < 2 * n / K + 4096, sizes
    # shares are pieces of SPACE (runs of Morton cells over the scene box): rays of one 16^3 cell never straddle two shares
    ext = np.where(hi > lo, hi - lo, 1.0)
    cell = np.clip(((rays[:, :3] - lo) / ext * 16).astype(np.int64), 0, 15)
    key = (cell[:, 0] * 16 + cell[:, 1]) * 16 + cell[:, 2]
    per_cell_owners = {}
    straddle = 0
    for c, k in zip(key[::97], owner[::97]):
        if per_cell_owners.setdefault(int(c), int(k)) != int(k):
            straddle += 1
    assert straddle <= len(key[::97]) * 0.02, straddle          # (the library's box is that of the triangles under the tree: edge cells may differ)
    sc.close()


def test_small_batches_and_no_accelerator_are_cut_by_index():
    import torch
    si = scenes.random_tris(500, seed=4)
    lo, hi = scenes.scene_bounds(si)
    for accel, n in ((api.ACCEL_BVH4, 10007), (api.ACCEL_NONE, 3001)):
        built = api.BuiltScene(si, accel)
        sc = api.Scene(built, 0)
        rays = scenes.random_rays(n, lo, hi, seed=5)
        d_rays = torch.from_numpy(rays).cuda()
        o = ol.OracleScene(si, accel)
        want = o.intersect(rays)
        o.close()
        K = 4
        sizes, owner = _check_closest(torch, sc, d_rays, n, K, want)
        bounds = [n * k // K for k in range(K + 1)]
        for k in range(K):
            assert np.all(owner[bounds[k]:bounds[k + 1]] == k)
        sc.close()


def test_shadow_shares_tile_the_batch():
    import torch
    si = scenes.courtyard(16, 32, 64, seed=7)
    built = api.BuiltScene(si, api.ACCEL_BVH4, sah=True)
    sc = api.Scene(built, 0)
    lo, hi = scenes.scene_bounds(si)
    n = 1 << 19
    rays = scenes.random_rays(n, lo, hi, seed=23)
    ld = (np.random.default_rng(3).random(n) * np.linalg.norm(hi - lo)).astype(np.float32)
    d_rays, d_ld = torch.from_numpy(rays).cuda(), torch.from_numpy(ld).cuda()
    t_full = torch.empty(n, dtype=torch.float32, device="cuda")
    occ_full = torch.empty(n, dtype=torch.uint8, device="cuda")
    sc.shadow_dev(d_rays.data_ptr(), d_ld.data_ptr(), n, t_full.data_ptr(), occ_full.data_ptr(), None)
    K = 5
    t_all = torch.full((n,), -3.0, dtype=torch.float32, device="cuda")
    occ_all = torch.full((n,), 9, dtype=torch.uint8, device="cuda")
    seen = 0
    for k in range(K):
        sc.shadow_dev_slice(d_rays.data_ptr(), d_ld.data_ptr(), n, t_all.data_ptr(), occ_all.data_ptr(), k, K, None)
        torch.cuda.synchronize()
        now = int((occ_all != 9).sum().item())
        assert now > seen
        seen = now
    assert seen == n
    assert ol.same_bits(t_all.cpu().numpy(), t_full.cpu().numpy()) and bool((occ_all == occ_full).all().item())
    sc.close()


def test_bad_slice_arguments():
    import torch
    si = scenes.random_tris(50, seed=1)
    built = api.BuiltScene(si, api.ACCEL_BVH4)
    sc = api.Scene(built, 0)
    d = torch.zeros((8, 6), dtype=torch.float32, device="cuda")
    h = torch.zeros((8, 5), dtype=torch.int32, device="cuda")
    for k, K in ((0, 0), (2, 2), (5, 1)):
        with pytest.raises(api.CrtError):
            sc.intersect_dev_slice(d.data_ptr(), 8, h.data_ptr(), k, K, None)
    sc.reserve_rays(1 << 20)
    sc.close()


def test_triangle_only_description_with_triangles_outside_the_tree():
    """validate_desc accepts a triangle-only description whose first objects are NOT under the accelerator
    (num_non_grid_obj > 0; the library's own builder never makes one).  intersect.h:403-411 tests them after the tree, strict `<`.
    A batch of <= 2^23 rays used to take the tail-splitting walk, whose settle step skips that loop."""
    import torch
    si = scenes.courtyard(16, 32, 64, seed=7)
    built = api.BuiltScene(si, api.ACCEL_BVH4, sah=True)
    d0 = built.desc
    lo, hi = scenes.scene_bounds(si)
    # three large triangles in front of everything else, appended to the triangle array, placed first in objects[]
    k = 3
    c = (lo + hi) * 0.5
    e = (hi - lo)
    extra = np.array([[[lo[0], c[1] + 0.1 * e[1] * j, lo[2]], [hi[0], c[1] + 0.1 * e[1] * j, lo[2]], [c[0], c[1] + 0.1 * e[1] * j, hi[2]]]
                      for j in range(k)], np.float32).reshape(k, 9)
    nt = d0.num_tris
    tri = np.concatenate([np.ctypeslib.as_array(C.cast(d0.tri_verts, C.POINTER(C.c_float)), (nt * 9,)).reshape(nt, 9), extra]).astype(np.float32)
    n0 = d0.num_obj
    obj_type = np.concatenate([np.full(k, api.TRIANGLE, np.int32), built.obj_type])
    obj_ref = np.concatenate([np.arange(nt, nt + k, dtype=np.int32), built.obj_ref])
    obj_id = np.concatenate([np.arange(n0, n0 + k, dtype=np.int32), built.obj_id])
    nodes = built.nodes4.copy()
    leaf = nodes["type"] >= 0
    nodes["child"][leaf] += k                                   # leaf children point INTO objects[]
    d = api.CrtSceneDesc()
    C.memmove(C.byref(d), C.byref(d0), C.sizeof(d))
    d.num_obj = n0 + k
    d.num_non_grid_obj = k
    d.obj_type, d.obj_ref, d.obj_id = obj_type.ctypes.data, obj_ref.ctypes.data, obj_id.ctypes.data
    d.tri_verts, d.num_tris = tri.ctypes.data, nt + k
    d.nodes4 = nodes.ctypes.data

    class Holder:
        desc = d
    sc = api.Scene(Holder, 0)
    o_h = ol.oracle().oracle_scene_from_desc(C.byref(d))
    assert o_h
    n = 1 << 19                                                  # binned, and small enough for the tail-splitting instance
    rays = scenes.random_rays(n, lo, hi, seed=31)
    d_rays = torch.from_numpy(rays).cuda()
    got = torch.empty((n, 5), dtype=torch.int32, device="cuda")
    sc.intersect_dev(d_rays.data_ptr(), n, got.data_ptr(), None)
    torch.cuda.synchronize()
    g = _hits(got)
    want = np.empty(n, api.HIT_DTYPE)
    cnt = api.CrtCounters()
    ol.oracle().oracle_intersect(o_h, rays.ctypes.data, n, want.ctypes.data, api.TRAVERSE_PRUNED, C.byref(cnt), 8)
    ol.oracle().oracle_scene_destroy(o_h)
    assert (want["obj"] >= n0).sum() > n // 20                    # the outside triangles win a good share of the rays
    assert ol.same_bits(g, want)
    sc.close()


@pytest.mark.parametrize("accel", [api.ACCEL_BVH4, api.ACCEL_GRID])
def test_launches_on_two_streams_overlap_without_touching_each_other(accel):
    """A scene holds two sets of launch scratch (queue head, ray order, stack overflow) and hands them out in turn, so launches
    enqueued on two streams run side by side.  Six launches of two different batches, alternating streams and sizes (binned and
    not), each into its own buffer: every one equals the launch made alone."""
    import torch
    si = scenes.courtyard(16, 32, 64, seed=7)
    built = api.BuiltScene(si, accel, sah=(accel == api.ACCEL_BVH4))
    sc = api.Scene(built, 0)
    lo, hi = scenes.scene_bounds(si)
    sizes = [(1 << 19) + 5, 70001, 1 << 19, 999, (1 << 18) + 64, 300000]
    batches = [torch.from_numpy(scenes.random_rays(m, lo, hi, seed=40 + i)).cuda() for i, m in enumerate(sizes)]
    alone = []
    for b in batches:
        h = torch.empty((b.shape[0], 5), dtype=torch.int32, device="cuda")
        sc.intersect_dev(b.data_ptr(), b.shape[0], h.data_ptr(), None)
        torch.cuda.synchronize()
        alone.append(h)
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    torch.cuda.synchronize()
    for rep in range(3):
        outs = [torch.full((b.shape[0], 5), SENT, dtype=torch.int32, device="cuda") for b in batches]
        torch.cuda.synchronize()
        for i, b in enumerate(batches):
            sc.intersect_dev(b.data_ptr(), b.shape[0], outs[i].data_ptr(), streams[i & 1].cuda_stream)
        torch.cuda.synchronize()
        for i in range(len(batches)):
            assert bool((outs[i] == alone[i]).all().item()), (rep, i)
    sc.close()
