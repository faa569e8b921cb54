"""Host-buffer entry points (crt_intersect_batch / crt_shadow_batch): the chunked, double-buffered
pipeline must give exactly what one device-resident launch gives, for pinned and for ordinary
buffers, across chunk boundaries; launches that share one crt_scene from several streams or host
threads are ordered by the library and must not disturb each other."""
import threading

import numpy as np
import pytest

from craytracer_amd import api, scenes
import oraclelib as ol

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def setup():
    si = scenes.courtyard(8, 32, 64, seed=4)
    sc = api.Scene(api.BuiltScene(si, api.ACCEL_BVH4), 0)
    lo, hi = scenes.scene_bounds(si)
    n = (2 << 22) + 12345                       # 3 chunks of 2^22, the last one ragged
    rays = scenes.random_rays(n, lo, hi, 17)
    yield si, sc, rays
    sc.close()


def dev_reference(sc, rays):
    import torch
    d_rays = torch.from_numpy(rays).cuda()
    d_hits = torch.empty((len(rays), 5), dtype=torch.int32, device="cuda")
    sc.intersect_dev(d_rays.data_ptr(), len(rays), d_hits.data_ptr(), None)
    torch.cuda.synchronize()
    return d_hits.cpu().numpy().view(api.HIT_DTYPE).reshape(-1)


def test_chunked_pipeline_equals_single_launch(setup):
    si, sc, rays = setup
    want = dev_reference(sc, rays)
    assert ol.same_bits(sc.intersect(rays), want)                    # pageable buffers
    hb_r = api.HostBuffer(rays.shape, np.float32)
    hb_h = api.HostBuffer(len(rays), api.HIT_DTYPE)
    hb_r.array[:] = rays
    hb_h.array.view(np.uint8)[:] = 0xAB
    got = sc.intersect(hb_r.array, out=hb_h.array)                   # pinned buffers
    assert ol.same_bits(got, want)
    m = 1 << 16
    o = ol.OracleScene(si, api.ACCEL_BVH4)
    assert ol.same_bits(want[:m], o.intersect(rays[:m]))
    hb_r.close(); hb_h.close()


def test_shadow_pipeline(setup):
    import torch
    si, sc, rays = setup
    n = len(rays)
    ld = (np.random.default_rng(3).random(n) * 6).astype(np.float32)
    d_rays = torch.from_numpy(rays).cuda()
    d_ld = torch.from_numpy(ld).cuda()
    d_t = torch.empty(n, dtype=torch.float32, device="cuda")
    d_occ = torch.empty(n, dtype=torch.uint8, device="cuda")
    sc.shadow_dev(d_rays.data_ptr(), d_ld.data_ptr(), n, d_t.data_ptr(), d_occ.data_ptr(), None)
    torch.cuda.synchronize()
    t, occ = sc.shadow(rays, ld)
    assert ol.same_bits(t, d_t.cpu().numpy())
    assert np.array_equal(occ, d_occ.cpu().numpy().astype(bool))


def test_launches_from_two_streams_are_ordered(setup):
    import torch
    si, sc, rays = setup
    n = 1 << 20
    want = dev_reference(sc, rays[: 2 * n])
    d_rays = torch.from_numpy(rays[: 2 * n]).cuda()
    d_hits = torch.zeros((2 * n, 5), dtype=torch.int32, device="cuda")
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    torch.cuda.synchronize()
    for _ in range(3):                                               # back to back, no host sync in between
        sc.intersect_dev(d_rays.data_ptr(), n, d_hits.data_ptr(), s1.cuda_stream)
        sc.intersect_dev(d_rays.data_ptr() + n * 24, n, d_hits.data_ptr() + n * 20, s2.cuda_stream)
    torch.cuda.synchronize()
    assert ol.same_bits(d_hits.cpu().numpy().view(api.HIT_DTYPE).reshape(-1), want)


def test_two_host_threads_share_a_scene(setup):
    si, sc, rays = setup
    n = 300000
    want = dev_reference(sc, rays[: 2 * n])
    out = [None, None]

    def work(k):
        for _ in range(2):
            out[k] = sc.intersect(rays[k * n:(k + 1) * n])

    th = [threading.Thread(target=work, args=(k,)) for k in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert ol.same_bits(out[0], want[:n]) and ol.same_bits(out[1], want[n:])


def test_ray_binning_changes_nothing_but_the_order(setup):
    """crt_scene_set_ray_binning: same bits with and without, closest and any-hit, on a batch large
    enough (>= 2^18 rays) for the binning step to run; rays outside the scene box and rays with
    NaN / inf origins must survive the cell computation."""
    import torch
    si, sc, rays = setup
    n = 1 << 20
    r = rays[:n].copy()
    r[:1000, :3] *= 50.0                      # far outside the box
    r[1000:1010, 0] = np.nan
    r[1010:1020, 1] = np.inf
    r[1020:1030, 2] = -np.inf
    ld = (np.random.default_rng(5).random(n) * 6).astype(np.float32)
    out = {}
    for on in (True, False):
        sc.set_ray_binning(on)
        d_rays = torch.from_numpy(r).cuda()
        d_ld = torch.from_numpy(ld).cuda()
        d_hits = torch.empty((n, 5), dtype=torch.int32, device="cuda")
        d_t = torch.empty(n, dtype=torch.float32, device="cuda")
        d_occ = torch.empty(n, dtype=torch.uint8, device="cuda")
        sc.intersect_dev(d_rays.data_ptr(), n, d_hits.data_ptr(), None)
        sc.shadow_dev(d_rays.data_ptr(), d_ld.data_ptr(), n, d_t.data_ptr(), d_occ.data_ptr(), None)
        torch.cuda.synchronize()
        out[on] = (d_hits.cpu().numpy(), d_t.cpu().numpy(), d_occ.cpu().numpy())
    sc.set_ray_binning(True)
    for a, b in zip(out[True], out[False]):
        assert ol.same_bits(a, b)
    o = ol.OracleScene(si, api.ACCEL_BVH4)
    assert ol.same_bits(out[True][0][:2048].view(api.HIT_DTYPE).reshape(-1), o.intersect(r[:2048]))
