"""crt_intersect_batch_dev_noalloc (the entry point for callers that cannot allocate) against
crt_intersect_batch_dev, through the C ABI, bit for bit: same kernels, the scratch memory in the library's
file-scope device array instead of the scene's own buffers.

Two calls in a row on one GPU, the second with a smaller input (the array keeps the first call's contents:
queue head, histogram, ray order, stack overflow), on the QBVH walk with the binning step in play (first
call) and below its threshold (second call), then the other accelerators, the size limit, and two scenes
taking turns on the one array.  This file is collected last on purpose (pytest -x stops at a failure)."""
import numpy as np
import pytest

from craytracer_amd import api, scenes
import oraclelib as ol

pytestmark = pytest.mark.gpu


def _dev_rays(torch, rays):
    return torch.from_numpy(np.ascontiguousarray(rays, np.float32)).cuda()


def _both(torch, sc, d_rays, n):
    want = torch.full((n, 5), -7, dtype=torch.int32, device="cuda")
    got = torch.full((n, 5), -9, dtype=torch.int32, device="cuda")
    sc.intersect_dev(d_rays.data_ptr(), n, want.data_ptr(), None)
    sc.intersect_dev_noalloc(d_rays.data_ptr(), n, got.data_ptr(), None)
    torch.cuda.synchronize()
    return (want.cpu().numpy().view(api.HIT_DTYPE).reshape(-1), got.cpu().numpy().view(api.HIT_DTYPE).reshape(-1))


def test_two_calls_second_smaller_qbvh():
    import torch
    si = scenes.courtyard(16, 32, 64, seed=7)                    # 65 548 triangles, depth > 5: the stack overflow area is used
    built = api.BuiltScene(si, api.ACCEL_BVH4)
    sc = api.Scene(built, 0)
    lo, hi = scenes.scene_bounds(si)
    n1, n2 = (1 << 19) + 777, 5000                               # >= 2^18: binned; ragged tail.  Then below the threshold
    rays1 = scenes.random_rays(n1, lo, hi, seed=11)
    rays2 = scenes.random_rays(n2, lo, hi, seed=12)
    d1, d2 = _dev_rays(torch, rays1), _dev_rays(torch, rays2)
    mem0 = torch.cuda.mem_get_info()[0]
    want1, got1 = _both(torch, sc, d1, n1)
    assert (want1["obj"] >= 0).sum() > n1 // 4                   # the scene is hit
    assert ol.same_bits(got1, want1)
    want2, got2 = _both(torch, sc, d2, n2)
    assert ol.same_bits(got2, want2)
    # and against the CPU oracle, so that "equal" cannot mean "equally wrong"
    o = ol.OracleScene(si, api.ACCEL_BVH4)
    assert ol.same_bits(got2, o.intersect(rays2))
    o.close()
    # the larger one again after the smaller: nothing of call 2 leaks into call 3
    got3 = torch.empty((n1, 5), dtype=torch.int32, device="cuda")
    sc.intersect_dev_noalloc(d1.data_ptr(), n1, got3.data_ptr(), None)
    torch.cuda.synchronize()
    assert ol.same_bits(got3.cpu().numpy().view(api.HIT_DTYPE).reshape(-1), want1)
    del got3
    # no device memory was taken by the calls (torch's own result tensors are released again)
    torch.cuda.empty_cache()
    assert torch.cuda.mem_get_info()[0] >= mem0 - (64 << 20)
    sc.close()


@pytest.mark.parametrize("accel", [api.ACCEL_NONE, api.ACCEL_GRID, api.ACCEL_BVH, api.ACCEL_BVH4])
def test_every_accelerator_and_shape_kind(accel):
    import torch
    si = scenes.analytic_scene()
    sc = api.Scene(api.BuiltScene(si, accel), 0)
    rays = scenes.random_rays(20000, np.float32([-4, -2, -4]), np.float32([4, 4, 4]), seed=3)
    d = _dev_rays(torch, rays)
    want, got = _both(torch, sc, d, len(rays))
    assert ol.same_bits(got, want)
    want, got = _both(torch, sc, d, 333)
    assert ol.same_bits(got, want)
    sc.close()


def test_size_limit_and_bad_arguments():
    import torch
    si = scenes.courtyard(4, 16, 32, seed=1)
    sc = api.Scene(api.BuiltScene(si, api.ACCEL_BVH4), 0)
    d = torch.zeros((8, 6), dtype=torch.float32, device="cuda")
    h = torch.zeros((8, 5), dtype=torch.int32, device="cuda")
    L = api.lib()
    assert L.crt_intersect_batch_dev_noalloc(sc._h, d.data_ptr(), api.NOALLOC_MAX_RAYS + 1, h.data_ptr(), None) == -5   # CRT_E_TOOBIG
    assert L.crt_intersect_batch_dev_noalloc(sc._h, None, 8, h.data_ptr(), None) == -1
    assert L.crt_intersect_batch_dev_noalloc(None, d.data_ptr(), 8, h.data_ptr(), None) == -1
    assert L.crt_intersect_batch_dev_noalloc(sc._h, d.data_ptr(), 0, h.data_ptr(), None) == 0
    sc.close()


def test_the_limit_itself_and_two_scenes_taking_turns():
    import torch
    si_a = scenes.courtyard(8, 32, 64, seed=2)
    si_b = scenes.buddha_like(64, 128, seed=4)
    sa = api.Scene(api.BuiltScene(si_a, api.ACCEL_BVH4, sah=True), 0)
    sb = api.Scene(api.BuiltScene(si_b, api.ACCEL_BVH4), 0)
    n = api.NOALLOC_MAX_RAYS                                     # the largest batch the array's ray order holds
    lo, hi = scenes.scene_bounds(si_a)
    d = _dev_rays(torch, scenes.random_rays(n, lo, hi, seed=21))
    want_a, got_a = _both(torch, sa, d, n)
    assert ol.same_bits(got_a, want_a)
    lo, hi = scenes.scene_bounds(si_b)
    d2 = _dev_rays(torch, scenes.random_rays(1 << 18, lo, hi, seed=22))
    want_b, got_b = _both(torch, sb, d2, 1 << 18)                # another scene on the same array, after a sync
    assert ol.same_bits(got_b, want_b)
    want_a2, got_a2 = _both(torch, sa, d, 4097)
    assert ol.same_bits(got_a2, want_a[:4097]) and ol.same_bits(want_a2, want_a[:4097])
    sa.close()
    sb.close()


def test_two_scenes_on_two_streams_without_a_sync_between():
    """every no-allocation launch on a GPU shares that GPU's array, whichever scene it comes through, so the library
    orders them per DEVICE (one library-owned event).  Two scenes, two streams, launches alternating with no host
    synchronisation in between: each output equals that scene's crt_intersect_batch_dev output."""
    import torch
    si_a = scenes.courtyard(8, 32, 64, seed=6)
    si_b = scenes.buddha_like(64, 128, seed=8)
    sa = api.Scene(api.BuiltScene(si_a, api.ACCEL_BVH4, sah=True), 0)
    sb = api.Scene(api.BuiltScene(si_b, api.ACCEL_BVH4), 0)
    n = 1 << 20
    da = _dev_rays(torch, scenes.random_rays(n, *scenes.scene_bounds(si_a), seed=31))
    db = _dev_rays(torch, scenes.random_rays(n, *scenes.scene_bounds(si_b), seed=32))
    want_a = torch.empty((n, 5), dtype=torch.int32, device="cuda")
    want_b = torch.empty((n, 5), dtype=torch.int32, device="cuda")
    sa.intersect_dev(da.data_ptr(), n, want_a.data_ptr(), None)
    sb.intersect_dev(db.data_ptr(), n, want_b.data_ptr(), None)
    torch.cuda.synchronize()
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    outs = []
    for k in range(6):
        ga = torch.full((n, 5), -9, dtype=torch.int32, device="cuda")
        gb = torch.full((n, 5), -9, dtype=torch.int32, device="cuda")
        outs.append((ga, gb))
    torch.cuda.synchronize()
    for ga, gb in outs:                                            # no synchronisation inside this loop
        sa.intersect_dev_noalloc(da.data_ptr(), n, ga.data_ptr(), s1.cuda_stream)
        sb.intersect_dev_noalloc(db.data_ptr(), n, gb.data_ptr(), s2.cuda_stream)
    torch.cuda.synchronize()
    for ga, gb in outs:
        assert torch.equal(ga, want_a) and torch.equal(gb, want_b)
    sa.close()
    sb.close()
