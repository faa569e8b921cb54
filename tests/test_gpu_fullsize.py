"""Parity at BASELINE.json's full sizes (C3: ~1M triangles, C5: ~10M triangles, QBVH, 2^22..2^24
incoherent rays), where the oracle cannot replay the whole batch in seconds.  Size-independent
properties of the domain are checked on every ray, and the oracle on a bounded sample:

 * self-consistency: every reported hit (t, beta, gamma) is bit-identical to calcTriangleIntersect
   (triangle.cpp:14-77) re-evaluated for (that ray, the reported triangle) in float32, op by op;
   every miss is (TMAX, -1, -1, 0, 0);
 * order independence: a permuted batch gives the permuted hits (rays are independent; the
   persistent queue / lane refill must not leak state between rays);
 * reference control flow (EXACT) == best-t pruned walk (PRUNED), bit for bit;
 * any-hit vs closest-hit: light_dist just below the closest t -> unoccluded (TMAX);
   just above -> occluded with shadow t < light_dist; a miss never occludes;
 * brute force: the accelerator-free linear loop over all triangles gives the same t;
 * the oracle (reference control flow, same tree) on the first 2^18 rays: bit-identical.
"""
import os
import sys

import numpy as np
import pytest

from craytracer_amd import api, scenes
import oraclelib as ol

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

CASES = {
    # name: (scene factory, rays, builder): the reference's median-split QBVH and the SAH-built one (CRT_BUILD_SAH)
    "buddha-1M": (lambda: scenes.buddha_like(512, 1024, seed=3), 1 << 22, False),
    "courtyard-10M": (lambda: scenes.courtyard(640, 64, 128, seed=5), 1 << 24, False),
    "buddha-1M@sah": (lambda: scenes.buddha_like(512, 1024, seed=3), 1 << 22, True),
    "courtyard-10M@sah": (lambda: scenes.courtyard(640, 64, 128, seed=5), 1 << 24, True),
}
TMAX = api.TMAX
K_EPSILON = api.K_EPSILON


def tri_eval(tv, rays):
    """calcTriangleIntersect (triangle.cpp:14-77) in numpy float32; each op rounds separately."""
    f32 = np.float32
    v0, v1, v2 = tv[:, 0], tv[:, 1], tv[:, 2]
    o, dr = rays[:, :3], rays[:, 3:]
    a, b, c, d = v0[:, 0] - v1[:, 0], v0[:, 0] - v2[:, 0], dr[:, 0], v0[:, 0] - o[:, 0]
    e, f, g, h = v0[:, 1] - v1[:, 1], v0[:, 1] - v2[:, 1], dr[:, 1], v0[:, 1] - o[:, 1]
    i, j, k, l = v0[:, 2] - v1[:, 2], v0[:, 2] - v2[:, 2], dr[:, 2], v0[:, 2] - o[:, 2]
    m, n, p = f * k - g * j, h * k - g * l, f * l - h * j
    q, s = g * i - e * k, e * j - f * i
    with np.errstate(all="ignore"):
        inv = f32(1.0) / (a * m + b * q + c * s)
        beta = (d * m - b * n - c * p) * inv
        r = e * l - h * i
        gamma = (a * n + d * q + c * r) * inv
        t = (a * p - b * r + d * s) * inv
        rej = (beta < 0) | (gamma < 0) | (beta + gamma > f32(1.0)) | (t < K_EPSILON)
    return np.where(rej, TMAX, t).astype(f32), beta.astype(f32), gamma.astype(f32)


@pytest.fixture(scope="module", params=sorted(CASES))
def case(request):
    import torch
    import bench
    make, n, sah = CASES[request.param]
    si = make()
    built = api.BuiltScene(si, api.ACCEL_BVH4, sah=sah)
    sc = api.Scene(built, 0)
    dev = torch.device("cuda", 0)
    d_rays = bench.gen_incoherent_rays(sc, si, n, 0xC0FFEE, dev)
    d_hits = torch.empty((n, 5), dtype=torch.int32, device=dev)
    sc.intersect_dev(d_rays.data_ptr(), n, d_hits.data_ptr(), None)
    torch.cuda.synchronize()
    hits = d_hits.cpu().numpy().view(api.HIT_DTYPE).reshape(-1)
    yield dict(name=request.param, si=si, built=built, sc=sc, d_rays=d_rays, rays=d_rays.cpu().numpy(), hits=hits, n=n)
    sc.close()


def test_hits_are_self_consistent(case):
    hits, rays = case["hits"], case["rays"]
    hit = hits["obj"] >= 0
    assert 0.5 < hit.mean() <= 1.0
    miss = hits[~hit]
    assert (miss["t"] == TMAX).all() and (miss["part"] == -1).all()
    assert (miss["beta"] == 0).all() and (miss["gamma"] == 0).all()
    tv = case["si"].tri_verts.reshape(-1, 3, 3)[hits["obj"][hit]]
    t, beta, gamma = tri_eval(tv, rays[hit])
    assert ol.same_bits(t, hits["t"][hit])
    assert ol.same_bits(beta, hits["beta"][hit]) and ol.same_bits(gamma, hits["gamma"][hit])
    assert (hits["part"][hit] == 0).all()


def test_oracle_on_a_sample(case):
    m = 1 << 18
    o = ol.OracleScene(case["si"], built=case["built"])
    ref = o.intersect(case["rays"][:m], mode=api.TRAVERSE_EXACT, nthreads=min(os.cpu_count() or 1, 16))
    assert ol.same_bits(case["hits"][:m], ref)


def test_order_independence(case):
    import torch
    n = case["n"]
    g = torch.Generator(device="cuda")
    g.manual_seed(77)
    perm = torch.randperm(n, generator=g, device="cuda")
    d_rays = case["d_rays"][perm].contiguous()
    d_hits = torch.empty((n, 5), dtype=torch.int32, device="cuda")
    case["sc"].intersect_dev(d_rays.data_ptr(), n, d_hits.data_ptr(), None)
    torch.cuda.synchronize()
    got = d_hits.cpu().numpy().view(api.HIT_DTYPE).reshape(-1)
    assert ol.same_bits(got, case["hits"][perm.cpu().numpy()])


def test_reference_flow_equals_pruned_walk(case):
    import torch
    n = min(case["n"], 1 << 22)
    sc = case["sc"]
    d_hits = torch.empty((n, 5), dtype=torch.int32, device="cuda")
    sc.set_traverse_mode(api.TRAVERSE_EXACT)
    try:
        sc.intersect_dev(case["d_rays"].data_ptr(), n, d_hits.data_ptr(), None)
        torch.cuda.synchronize()
    finally:
        sc.set_traverse_mode(api.TRAVERSE_AUTO)
    got = d_hits.cpu().numpy().view(api.HIT_DTYPE).reshape(-1)
    assert ol.same_bits(got, case["hits"][:n])


def test_any_hit_agrees_with_closest_hit(case):
    import torch
    n = min(case["n"], 1 << 22)
    sc, hits = case["sc"], case["hits"][:n]
    hit = hits["obj"] >= 0
    t = hits["t"].astype(np.float64)
    below = np.where(hit, t * (1 - 1e-4) - 1e-5, 1e30).astype(np.float32)
    above = np.where(hit, t * (1 + 1e-4) + 1e-4, 1e30).astype(np.float32)
    d_t = torch.empty(n, dtype=torch.float32, device="cuda")
    d_occ = torch.empty(n, dtype=torch.uint8, device="cuda")
    for ld, expect_occ in ((below, False), (above, True)):
        d_ld = torch.from_numpy(ld).cuda()
        sc.shadow_dev(case["d_rays"].data_ptr(), d_ld.data_ptr(), n, d_t.data_ptr(), d_occ.data_ptr(), None)
        torch.cuda.synchronize()
        st, occ = d_t.cpu().numpy(), d_occ.cpu().numpy().astype(bool)
        assert (st[~hit] == TMAX).all() and not occ[~hit].any()
        if expect_occ:
            assert (st[hit] < ld[hit]).all() and (st[hit] >= hits["t"][hit]).all()
            # the caller's test (trace.h:348); a first-found hit inside [ld - eps, ld) is class T5
            assert np.array_equal(occ[hit], st[hit] < (ld[hit] - K_EPSILON))
            assert occ[hit].mean() > 0.999
        else:
            assert (st[hit] == TMAX).all() and not occ[hit].any()


def test_brute_force_sample(case):
    m = 2048
    none = api.Scene(api.BuiltScene(case["si"], api.ACCEL_NONE), 0)
    brute = none.intersect(case["rays"][:m])
    none.close()
    got = case["hits"][:m]
    assert ol.same_bits(brute["t"], got["t"])
    differ = brute["obj"] != got["obj"]          # only exact-t ties (shared edges) may name another triangle
    assert differ.mean() < 1e-3
    assert (brute["t"][differ] == got["t"][differ]).all()


def test_any_hit_oracle_on_a_sample(case):
    """Shadow rays from the hit points toward a ceiling light (bench.gen_shadow_rays): the first-found
    t of the reference's walk order (class T5) and the caller's boolean, bit-identical to the oracle,
    and the kernel's node / primitive counters equal to the oracle's."""
    import torch
    import bench
    m = 1 << 18
    sc = case["sc"]
    d_rays, d_ld = bench.gen_shadow_rays(sc, case["si"], case["d_rays"][:m].contiguous(), 99, torch.device("cuda", 0))
    d_t = torch.empty(m, dtype=torch.float32, device="cuda")
    d_occ = torch.empty(m, dtype=torch.uint8, device="cuda")
    o = ol.OracleScene(case["si"], built=case["built"])
    t, occ, oc = o.shadow(d_rays.cpu().numpy(), d_ld.cpu().numpy(), nthreads=min(os.cpu_count() or 1, 16), counters=True)
    assert 0.02 < occ.mean() < 0.98
    # 128-byte nodes: the oracle's walk step for step; 64-byte conservative copy (the default): a superset of it, same results
    for compressed in (False, True):
        sc.set_node_compression(compressed)
        d_t.fill_(-1.0)
        d_occ.fill_(7)
        sc.enable_counters(True)
        sc.reset_counters()
        sc.shadow_dev(d_rays.data_ptr(), d_ld.data_ptr(), m, d_t.data_ptr(), d_occ.data_ptr(), None)
        cnt = sc.counters()
        sc.enable_counters(False)
        assert ol.same_bits(d_t.cpu().numpy(), t), compressed
        assert np.array_equal(d_occ.cpu().numpy().astype(bool), occ), compressed
        if compressed:
            assert oc["nodes_visited"] <= cnt["nodes_visited"] <= 1.1 * oc["nodes_visited"]
            assert oc["prims_tested"] <= cnt["prims_tested"] <= 1.1 * oc["prims_tested"]
        else:
            assert cnt["nodes_visited"] == oc["nodes_visited"] and cnt["prims_tested"] == oc["prims_tested"]


@pytest.mark.parametrize("accel", [api.ACCEL_BVH, api.ACCEL_GRID])
def test_other_accelerators_at_1m_triangles(accel):
    """Binary BVH and uniform grid on the ~1M-triangle scene, 2^20 incoherent rays: the oracle on a
    sample (pruned walk for the BVH on 2^17 rays, the reference's full walk on 2^13), every ray's t
    against the QBVH result (same triangles, same intersector: t is accelerator-independent; ids may
    differ only on exact-t ties, where the BVH keeps the later and the others the earlier object)."""
    import torch
    import bench
    si = scenes.buddha_like(512, 1024, seed=3)
    n = 1 << 20
    dev = torch.device("cuda", 0)
    q = api.Scene(api.BuiltScene(si, api.ACCEL_BVH4), 0)
    d_rays = bench.gen_incoherent_rays(q, si, n, 0xC0FFEE, dev)
    rays = d_rays.cpu().numpy()
    ref4 = q.intersect(rays)
    q.close()
    built = api.BuiltScene(si, accel)
    sc = api.Scene(built, 0)
    hits = sc.intersect(rays)
    ld = np.where(ref4["obj"] >= 0, ref4["t"] * np.float32(1.5), np.float32(10.0)).astype(np.float32)
    st, occ = sc.shadow(rays, ld)
    sc.close()
    nz = (rays[:, 3:] != 0).all(1)                    # the grid's NaN flow (class T2) is checked against the oracle below
    assert ol.same_bits(hits["t"][nz], ref4["t"][nz])
    differ = hits["obj"] != ref4["obj"]
    assert differ.mean() < 1e-3 and (hits["t"][differ] == ref4["t"][differ]).all()
    o = ol.OracleScene(si, accel, built=built)
    threads = min(os.cpu_count() or 1, 16)
    m = 1 << 17
    if accel == api.ACCEL_BVH:
        assert ol.same_bits(hits[:m], o.intersect(rays[:m], mode=api.TRAVERSE_PRUNED, nthreads=threads))
        m = 1 << 13
    assert ol.same_bits(hits[:m], o.intersect(rays[:m], mode=api.TRAVERSE_EXACT, nthreads=threads))
    wt, wocc = o.shadow(rays[:m], ld[:m], nthreads=threads)
    assert ol.same_bits(st[:m], wt) and np.array_equal(occ[:m], wocc)


@pytest.mark.parametrize("accel", [api.ACCEL_BVH4, api.ACCEL_BVH])
def test_mesh_with_lights_and_spheres_takes_the_pruned_path(accel):
    """~1M triangles plus rectangles (area light), spheres, a disk and a ground plane: AUTO must
    resolve to the pruned walk (all boxes bound their shapes) and that walk must equal the
    reference's control flow bit for bit -- on 2^20 incoherent rays GPU vs GPU, and against the
    oracle in the reference's control flow on a sample."""
    import torch
    si = scenes.lit_mesh_scene()
    built = api.BuiltScene(si, accel)
    sc = api.Scene(built, 0)
    assert sc.traverse_mode() == api.TRAVERSE_PRUNED
    n = 1 << 20
    lo, hi = np.zeros(3, np.float32), np.ones(3, np.float32)
    prim = scenes.random_rays(n, lo, hi, seed=23)
    h0 = sc.intersect(prim)
    ok = h0["obj"] >= 0
    p = prim[:, :3] + prim[:, 3:] * np.where(ok, h0["t"], 0)[:, None]
    d = np.random.default_rng(5).normal(size=(n, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    rays = np.concatenate([p, d], 1).astype(np.float32)              # origins ON surfaces, random directions
    pruned = sc.intersect(rays)
    ld = np.full(n, 0.6, np.float32)
    st_p, occ_p = sc.shadow(rays, ld)
    if accel == api.ACCEL_BVH4 or True:
        sc.set_traverse_mode(api.TRAVERSE_EXACT)
        exact = sc.intersect(rays)
        sc.set_traverse_mode(api.TRAVERSE_AUTO)
        bad = pruned.view(np.uint8).reshape(n, -1) != exact.view(np.uint8).reshape(n, -1)
        assert not bad.any(), (int(bad.any(1).sum()), pruned[bad.any(1)][:4], exact[bad.any(1)][:4])
    types = si.obj_type[pruned["obj"][pruned["obj"] >= 0]]
    assert (types == api.RECTANGLE).sum() > 1000 and (types == api.SPHERE).sum() > 1000    # the shapes do get hit
    m = 1 << 14
    o = ol.OracleScene(si, accel, built=built)
    want = o.intersect(rays[:m], mode=api.TRAVERSE_EXACT, nthreads=min(os.cpu_count() or 1, 16))
    same_id = (pruned["obj"][:m] == want["obj"]) & (pruned["part"][:m] == want["part"])
    dt = np.abs(pruned["t"][:m].astype(np.float64) - want["t"]) <= 1e-5 * np.maximum(1.0, np.abs(want["t"]))
    assert (~(same_id & (dt | (pruned["t"][:m] == want["t"])))).sum() <= 4        # sphere seams (class T4)
    wt, wocc = o.shadow(rays[:m], ld[:m], nthreads=min(os.cpu_count() or 1, 16))
    assert (st_p[:m] != wt).sum() <= 4 and (occ_p[:m] != wocc).sum() <= 4
    sc.close()


@pytest.mark.parametrize("reserve", [False, True])
def test_batches_larger_than_the_presized_ray_order(reserve):
    """A batch above 2^24 rays: the ray-order buffers of the binning step are created for 2^24 and either grow inside the first such
    launch (a stream synchronise, hipFree, hipMalloc) or were sized ahead with crt_scene_reserve_rays.  Either way every record equals
    the one the same ray gets in a batch of ordinary size (rays are independent), on every accelerator's binned path, closest hit
    and any hit, and the oracle agrees on a sample."""
    import torch
    import bench
    si = scenes.courtyard(4, 64, 128, seed=5)
    n = (1 << 24) + 12345
    dev = torch.device("cuda", 0)
    for accel in (api.ACCEL_BVH4, api.ACCEL_GRID):
        built = api.BuiltScene(si, accel, sah=(accel == api.ACCEL_BVH4))
        sc = api.Scene(built, 0)
        if reserve:
            sc.reserve_rays(n)
        d_rays = bench.gen_incoherent_rays(sc, si, 1 << 22, 0xC0FFEE, dev).repeat(5, 1)[:n].contiguous()
        d_rays[1 << 22:, :3] += 1.0e-3 * torch.randn((n - (1 << 22), 3), device=dev, generator=torch.Generator(device=dev).manual_seed(5))
        d_hits = torch.empty((n, 5), dtype=torch.int32, device=dev)
        sc.intersect_dev(d_rays.data_ptr(), n, d_hits.data_ptr(), None)
        torch.cuda.synchronize()
        parts = torch.empty((n, 5), dtype=torch.int32, device=dev)
        for lo in range(0, n, 1 << 23):
            m = min(1 << 23, n - lo)
            sc.intersect_dev(d_rays.data_ptr() + lo * 24, m, parts.data_ptr() + lo * 20, None)
        torch.cuda.synchronize()
        assert bool((parts == d_hits).all().item())
        ld = torch.full((n,), 0.35, dtype=torch.float32, device=dev)
        t_a = torch.empty(n, dtype=torch.float32, device=dev)
        occ_a = torch.empty(n, dtype=torch.uint8, device=dev)
        t_b, occ_b = torch.empty_like(t_a), torch.empty_like(occ_a)
        sc.shadow_dev(d_rays.data_ptr(), ld.data_ptr(), n, t_a.data_ptr(), occ_a.data_ptr(), None)
        for lo in range(0, n, 1 << 23):
            m = min(1 << 23, n - lo)
            sc.shadow_dev(d_rays.data_ptr() + lo * 24, ld.data_ptr() + lo * 4, m, t_b.data_ptr() + lo * 4, occ_b.data_ptr() + lo, None)
        torch.cuda.synchronize()
        assert bool((t_a.view(torch.int32) == t_b.view(torch.int32)).all().item()) and bool((occ_a == occ_b).all().item())
        m = 1 << 16
        o = ol.OracleScene(si, accel, built=built)
        tail = d_rays[n - m:].cpu().numpy()
        ref = o.intersect(tail, nthreads=min(os.cpu_count() or 1, 16))
        assert ol.same_bits(d_hits[n - m:].cpu().numpy().view(api.HIT_DTYPE).reshape(-1), ref)
        o.close()
        sc.close()
        built.close()
