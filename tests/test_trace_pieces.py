"""Pieces of the shading path that DO compile from the reference's sources (oracle/_ref): the ShadeRec
an intersector leaves behind, the sampling maps, getVec3InLocalBasis, and the one-lobe Lambertian BSDF
(BSDF_addLambertian + BSDF_sample_f + BSDF_f, reflection.cpp).  The restatement in
oracle/crt_oracle_trace.cpp must give the same bits.  (pathTrace / estimateDirect themselves live in
trace.h, which needs GLFW: not covered here, "parity unpinned".)"""
import ctypes as C

import numpy as np
import pytest

from craytracer_amd import api, scenes
import oraclelib as ol

pytestmark = pytest.mark.ref


def _p(a):
    return a.ctypes.data


def test_shade_rec_of_triangle_rectangle_sphere():
    si, desc, pos, look = scenes.render_room(6, 12)
    rs = ol.RefScene(si, api.ACCEL_NONE)
    order, _ = rs.order()
    final_of = {int(i): k for k, i in enumerate(order)}
    cin = si.c_struct()
    rng = np.random.default_rng(4)
    rays = np.zeros((6000, 6), np.float32)
    rays[:, :3] = rng.random((6000, 3), dtype=np.float32) * 0.9 + 0.05
    d = rng.standard_normal((6000, 3)).astype(np.float32)
    rays[:, 3:] = d / np.linalg.norm(d, axis=1, keepdims=True)
    hits = rs.intersect(rays)
    seen = {}
    marker = np.float32(-123.25)
    L, R = ol.oracle(), ol.ref()
    for ray, h in zip(rays, hits):
        obj = int(h["obj"])
        if obj < 0:
            continue
        typ = int(si.obj_type[obj])
        if typ == api.DISK:                                    # emissive only in this form; its ShadeRec is never read
            continue
        want = np.zeros(12, np.float32)
        t = R.ref_shade_rec(rs.h, final_of[obj], _p(ray), _p(want), marker)
        assert np.float32(t) == h["t"]
        got = np.zeros(12, np.float32)
        L.oracle_trace_shade_rec(C.byref(cin), obj, _p(ray), h["t"], _p(got))
        assert ol.same_bits(got[:9], want[:9]), (typ, got, want)
        if typ == api.TRIANGLE:
            assert (want[9:] == marker).all()                 # triangle.cpp:154-168 never writes dpdu
        else:
            assert ol.same_bits(got[9:], want[9:]), (typ, got, want)
        seen[typ] = seen.get(typ, 0) + 1
    assert seen.get(api.TRIANGLE, 0) > 1000 and seen.get(api.RECTANGLE, 0) > 20 and seen.get(api.SPHERE, 0) > 50


def test_sampling_maps_and_local_basis():
    L, R = ol.oracle(), ol.ref()
    rng = np.random.default_rng(5)
    smp = rng.random((20000, 2), dtype=np.float32)
    smp[:8] = [[0, 0], [0, 1], [0.25, 1], [0.5, 1], [0.75, 1], [0.999999, 1], [0.5, 0], [0.125, 0.5]]
    nrm = rng.standard_normal((20000, 3)).astype(np.float32)
    nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    a2, b2, a3, b3, a4, b4 = (np.zeros(k, np.float32) for k in (2, 2, 3, 3, 3, 3))
    for s, n in zip(smp, nrm):
        L.oracle_trace_disk(_p(s), _p(a2)); R.ref_map_sample_to_disk(_p(s), _p(b2))
        L.oracle_trace_hemisphere(_p(s), _p(a3)); R.ref_map_sample_to_hemisphere(_p(s), _p(b3))
        assert ol.same_bits(a2, b2) and ol.same_bits(a3, b3)
        L.oracle_trace_local_basis(_p(a3), _p(n), _p(a4)); R.ref_vec3_in_local_basis(_p(b3), _p(n), _p(b4))
        assert ol.same_bits(a4, b4)


def test_lambertian_bsdf():
    L, R = ol.oracle(), ol.ref()
    rng = np.random.default_rng(6)
    for k in range(5000):
        n = rng.standard_normal(3).astype(np.float32); n /= np.linalg.norm(n)
        t = np.cross(n, rng.standard_normal(3)).astype(np.float32); t /= np.linalg.norm(t)
        b = np.cross(n, t).astype(np.float32)
        basis = np.concatenate([t, b, n]).astype(np.float32)
        wo = rng.standard_normal(3).astype(np.float32); wo /= np.linalg.norm(wo)
        if k % 50 == 0:
            wo = t.copy()                                      # grazing: wo_local.z == 0 up to rounding
        cd = rng.random(3, dtype=np.float32)
        smp = rng.random(2, dtype=np.float32)
        fa, wa, ea, fb, wb, eb = (np.zeros(3, np.float32) for _ in range(6))
        pa = L.oracle_trace_lambertian(_p(cd), _p(basis), _p(wo), _p(smp), _p(fa), _p(wa), _p(ea))
        pb = R.ref_lambertian_bsdf(_p(cd), _p(basis), _p(wo), _p(smp), _p(fb), _p(wb), _p(eb))
        assert np.float32(pa) == np.float32(pb)
        if pb != 0.0:
            assert ol.same_bits(fa, fb) and ol.same_bits(wa, wb)
        assert ol.same_bits(ea, eb)


def test_mirror_bsdf():
    L, R = ol.oracle(), ol.ref()
    rng = np.random.default_rng(7)
    for k in range(3000):
        n = rng.standard_normal(3).astype(np.float32); n /= np.linalg.norm(n)
        t = np.cross(n, rng.standard_normal(3)).astype(np.float32); t /= np.linalg.norm(t)
        b = np.cross(n, t).astype(np.float32)
        basis = np.concatenate([t, b, n]).astype(np.float32)
        wo = rng.standard_normal(3).astype(np.float32); wo /= np.linalg.norm(wo)
        cr = rng.random(3, dtype=np.float32)
        fa, wa, fb, wb = (np.zeros(3, np.float32) for _ in range(4))
        ga, gb = C.c_int(), C.c_int()
        pa = L.oracle_trace_mirror(_p(cr), _p(basis), _p(wo), _p(fa), _p(wa), C.byref(ga))
        pb = R.ref_mirror_bsdf(_p(cr), _p(basis), _p(wo), _p(fb), _p(wb), C.byref(gb))
        assert np.float32(pa) == np.float32(pb) == 1.0 and ga.value == gb.value == 17      # REFLECTION | SPECULAR
        assert ol.same_bits(fa, fb) and ol.same_bits(wa, wb)


def test_transparent_bsdf():
    """The `thin` SpecularTransmission lobe; its rand() is replayed: srand(seed) in the harness, the same
    srand + rand through libc here."""
    L, R = ol.oracle(), ol.ref()
    libc = C.CDLL(None)
    libc.rand.restype = C.c_int
    RAND_MAX = 2147483647
    rng = np.random.default_rng(8)
    n_reflect = 0
    for k in range(4000):
        n = rng.standard_normal(3).astype(np.float32); n /= np.linalg.norm(n)
        t = np.cross(n, rng.standard_normal(3)).astype(np.float32); t /= np.linalg.norm(t)
        b = np.cross(n, t).astype(np.float32)
        basis = np.concatenate([t, b, n]).astype(np.float32)
        wo = rng.standard_normal(3).astype(np.float32); wo /= np.linalg.norm(wo)
        if k % 4 == 0:                                         # grazing: large Fresnel reflectance
            wo = (t * np.float32(0.995) + n * np.float32(0.0999) * (1 if k % 8 else -1)).astype(np.float32)
        ior_in, ior_out = np.float32(1.0 + rng.random()), np.float32(1.0 + 0.6 * rng.random())
        seed = int(rng.integers(1, 2**31))
        libc.srand(seed)
        rand_float = np.float32(np.float32(libc.rand()) / np.float32(RAND_MAX))
        fa, wa, fb, wb = (np.zeros(3, np.float32) for _ in range(4))
        ga, gb = C.c_int(), C.c_int()
        pa = L.oracle_trace_transparent(ior_in, ior_out, rand_float, _p(basis), _p(wo), _p(fa), _p(wa), C.byref(ga))
        pb = R.ref_transparent_bsdf(ior_in, ior_out, seed, _p(basis), _p(wo), _p(fb), _p(wb), C.byref(gb))
        assert np.float32(pa) == np.float32(pb) and ga.value == gb.value == 19             # REFLECTION | TRANSMISSION | SPECULAR
        if pb != 0.0:
            assert ol.same_bits(fa, fb) and ol.same_bits(wa, wb)
        n_reflect += int(np.dot(wa, n) * np.dot(wo, n) > 0)
    assert 100 < n_reflect < 3900


def test_orennayar_bsdf():
    """MATTE with sigma != 0.  BSDF_f evaluates the lobe on the vectors it is given -- the reference gives it
    world-space ones (reflection.cpp:731) -- so the evaluation is compared on arbitrary unit vectors."""
    L, R = ol.oracle(), ol.ref()
    rng = np.random.default_rng(9)
    for k in range(5000):
        n = rng.standard_normal(3).astype(np.float32); n /= np.linalg.norm(n)
        t = np.cross(n, rng.standard_normal(3)).astype(np.float32); t /= np.linalg.norm(t)
        b = np.cross(n, t).astype(np.float32)
        basis = np.concatenate([t, b, n]).astype(np.float32)
        wo = rng.standard_normal(3).astype(np.float32); wo /= np.linalg.norm(wo)
        we = rng.standard_normal(3).astype(np.float32); we /= np.linalg.norm(we)
        if k % 25 == 0:
            we = np.array([0.0, 0.0, 1.0], np.float32)          # sin_theta == 0 branch of cosPhi / sinPhi
        rr = rng.random(3, dtype=np.float32)
        sigma = np.float32(rng.random() * 60.0 + 0.5)
        smp = rng.random(2, dtype=np.float32)
        fa, wa, ea, fb, wb, eb = (np.zeros(3, np.float32) for _ in range(6))
        pa = L.oracle_trace_orennayar(_p(rr), sigma, _p(basis), _p(wo), _p(smp), _p(we), _p(fa), _p(wa), _p(ea))
        pb = R.ref_orennayar_bsdf(_p(rr), sigma, _p(basis), _p(wo), _p(smp), _p(we), _p(fb), _p(wb), _p(eb))
        assert np.float32(pa) == np.float32(pb)
        if pb != 0.0:
            assert ol.same_bits(fa, fb) and ol.same_bits(wa, wb)
        assert ol.same_bits(ea, eb)


def test_bake_mesh_against_the_reference_matrix_steps():
    """crt_bake_mesh (product, host) against generateMeshTriangles' steps run through the reference's own eulerAngToMat3 /
    mat3_* / calcTriangleNormal: world vertices, normal_mat, FlatTriangle.normal -- bit for bit."""
    R = ol.ref()
    rng = np.random.default_rng(10)
    for k in range(300):
        pos = (rng.standard_normal((3, 3)) * (10.0 ** rng.integers(-2, 3))).astype(np.float32)
        scaling = (0.1 + 3.0 * rng.random(3)).astype(np.float32)
        orient = (rng.random(3) * 720.0 - 360.0).astype(np.float32)
        loc = (rng.standard_normal(3) * 50.0).astype(np.float32)
        if k % 10 == 0:
            orient[:] = [0.0, 90.0, -180.0]
        got = api.bake_mesh(pos, [0, 1, 2], scaling, orient, loc)
        wv, nm, fn = np.zeros(9, np.float32), np.zeros(9, np.float32), np.zeros(3, np.float32)
        R.ref_bake_triangle(_p(scaling), _p(orient), _p(loc), _p(pos), _p(wv), _p(nm), _p(fn))
        assert ol.same_bits(got["tri_verts"][0], wv) and ol.same_bits(got["normal_mat"], nm) and ol.same_bits(got["flat_normal"][0], fn)


def test_shade_rec_of_smooth_and_flat_mesh_triangles():
    """fillShadeRec of the oracle for SMOOTH_TRIANGLE / FLAT_TRIANGLE against rayIntersectSmoothTriangle /
    rayIntersectFlatTriangle themselves (a three-vertex Mesh built in the harness)."""
    L, R = ol.oracle(), ol.ref()
    rng = np.random.default_rng(11)
    marker = np.float32(-77.5)
    n_hit = 0
    for k in range(3000):
        v = rng.standard_normal(9).astype(np.float32)
        vn = rng.standard_normal((3, 3)).astype(np.float32); vn /= np.linalg.norm(vn, axis=1, keepdims=True)
        vt = rng.standard_normal((3, 3)).astype(np.float32); vt /= np.linalg.norm(vt, axis=1, keepdims=True)
        nm = (np.eye(3) + 0.4 * rng.standard_normal((3, 3))).astype(np.float32).reshape(9)
        fn = rng.standard_normal(3).astype(np.float32); fn /= np.linalg.norm(fn)
        centre = v.reshape(3, 3).mean(0)
        o = (centre + rng.standard_normal(3) * 3.0).astype(np.float32)
        target = (v.reshape(3, 3) * rng.dirichlet([1, 1, 1])[:, None]).sum(0)
        d = (target - o); d = (d / np.linalg.norm(d)).astype(np.float32)
        ray = np.concatenate([o, d]).astype(np.float32)
        for smooth in (1, 0):
            want, bg = np.zeros(12, np.float32), np.zeros(2, np.float32)
            t = R.ref_mesh_triangle_shade_rec(smooth, _p(v), _p(vn), _p(vt), _p(fn), _p(nm), _p(ray), _p(want), _p(bg), marker)
            if not t < 3.0e38:
                continue
            n_hit += 1
            si = api.SceneInput(v.reshape(1, 9), None, np.array([api.SMOOTH_TRIANGLE if smooth else api.FLAT_TRIANGLE], np.int32),
                                np.array([0], np.int32))
            ms = api.MeshShading(fn.reshape(1, 3), vn.reshape(1, 9), vt.reshape(1, 9), np.zeros(1, np.int32), nm.reshape(1, 9))
            cin = si.c_struct()
            got = np.zeros(12, np.float32)
            L.oracle_trace_shade_rec_mesh(C.byref(cin), C.byref(ms.c), 0, _p(ray), np.float32(t), bg[0], bg[1], _p(got))
            assert ol.same_bits(got[:9], want[:9]), (smooth, got, want)
            if smooth:
                assert ol.same_bits(got[9:], want[9:])
            else:
                assert (want[9:] == marker).all()               # rayIntersectFlatTriangle never writes dpdu either
    assert n_hit > 3000


def _ulps(a, b):
    """distance in float32 ulps (same sign assumed where it matters), elementwise"""
    a = np.asarray(a, np.float32).reshape(-1); b = np.asarray(b, np.float32).reshape(-1)
    return np.abs(a.view(np.int32).astype(np.int64) - b.view(np.int32).astype(np.int64))


def _rand_frame(rng):
    n = rng.standard_normal(3).astype(np.float32); n /= np.linalg.norm(n)
    t = np.cross(n, rng.standard_normal(3)).astype(np.float32); t /= np.linalg.norm(t)
    b = np.cross(n, t).astype(np.float32)
    return np.concatenate([t, b, n]).astype(np.float32)


def _unit(rng):
    v = rng.standard_normal(3).astype(np.float32)
    return (v / np.linalg.norm(v)).astype(np.float32)


def test_beckmann_pieces():
    """MicrofacetDistribution_D / _G / Lambda / _sample_wh (microfacet.cpp, BECKMANN), isotropic and anisotropic.  The
    restatement rounds exp / log / tan / atan once from double where the reference calls glibc's float routines, so a
    last-ulp difference is allowed where those enter (D, sample_wh); G and Lambda use none of them and must be identical."""
    L, R = ol.oracle(), ol.ref()
    rng = np.random.default_rng(21)
    worst, differing, n = 0, 0, 6000
    for k in range(n):
        ax = np.float32(0.02 + rng.random() * 0.8)
        ay = ax if k % 2 == 0 else np.float32(0.02 + rng.random() * 0.8)
        wo, wi, wh = _unit(rng), _unit(rng), _unit(rng)
        if k % 50 == 0:
            wh = np.array([1.0, 0.0, 0.0], np.float32)           # tan^2 theta infinite -> D = 0
        smp = rng.random(2, dtype=np.float32)
        a, b = np.zeros(6, np.float32), np.zeros(6, np.float32)
        L.oracle_trace_beckmann(ax, ay, _p(wo), _p(wi), _p(wh), _p(smp), _p(a))
        R.ref_beckmann(ax, ay, _p(wo), _p(wi), _p(wh), _p(smp), _p(b))
        assert ol.same_bits(a[1:2], b[1:2])                      # G
        assert abs(float(a[2]) - float(b[2])) <= 3e-7 * max(1.0, abs(float(b[2])))   # Lambda, recovered from the reference's G1 as 1/G1 - 1
        # D, and the direction: a 1-ulp change of log / tan moves a component of wh by a few ulps of the largest one
        assert _ulps(a[0], b[0]).max() <= 2, (a[0], b[0])
        assert np.abs(a[3:] - b[3:]).max() <= 4e-6, (a, b)    # sin_theta = sqrt(1 - cos^2) amplifies the ulp when cos_theta is near 1
        u = int(max(_ulps(a[0], b[0]).max(), (a[3:] != b[3:]).any()))
        differing += u > 0
    assert differing < 0.02 * n, differing


@pytest.mark.parametrize("kind", ["PLASTIC", "METAL", "GLASS"])
def test_glossy_material_bsdfs(kind):
    """The BSDFs of PLASTIC (FresnelBlendDiffuse + FresnelBlendSpecular: two lobes, so the lobe pick, the remapped sample
    and the added pdf / f of the other lobe are exercised), METAL (MicrofacetReflection, conductor Fresnel) and GLASS
    (MicrofacetFresnel: reflection or refraction picked by rand(), replayed through srand) against the reference's
    reflection.cpp / microfacet.cpp: flags identical, pdf / f / wi within the last-ulp allowance of the transcendental
    functions (see test_beckmann_pieces), the fraction of results that are not bit-identical reported and bounded;
    BSDF_f as estimateDirect calls it (world vectors, glossy lobes excluded) identical."""
    L, R = ol.oracle(), ol.ref()
    rng = np.random.default_rng({"PLASTIC": 31, "METAL": 32, "GLASS": 33}[kind])
    n, inexact, zero_pdf, refl, refr = 4000, 0, 0, 0, 0
    for k in range(n):
        basis = _rand_frame(rng)
        wo, we = _unit(rng), _unit(rng)
        smp = rng.random(2, dtype=np.float32)
        ur = float(np.float32(0.05 + rng.random() * 0.6))
        vr = ur if k % 3 else float(np.float32(0.05 + rng.random() * 0.6))
        if kind == "PLASTIC":
            m = api.material(api.MAT_PLASTIC, rng.random(3), ks=rng.random(3) * 0.6, uroughness=ur)
        elif kind == "METAL":
            eta, kk = list(api.METALS.values())[k % len(api.METALS)]
            m = api.material(api.MAT_METAL, eta, k=kk, uroughness=ur, vroughness=vr)
        else:
            m = api.material(api.MAT_GLASS, (1, 1, 1), ior_in=1.5 if k % 4 else 1.33, ior_out=1.0, uroughness=ur, vroughness=vr)
        marr = np.array([m], api.MATERIAL_DTYPE)
        seed = 1000 + k
        u = R.ref_rand_float_after_srand(seed)
        fa, wa, ea, fb, wb, eb = (np.zeros(3, np.float32) for _ in range(6))
        ga, gb = C.c_int(0), C.c_int(0)
        pa = L.oracle_trace_material_bsdf(_p(marr), u, _p(basis), _p(wo), _p(smp), _p(we), _p(fa), _p(wa), C.byref(ga), _p(ea))
        pb = R.ref_material_bsdf(_p(marr), seed, _p(basis), _p(wo), _p(smp), _p(we), _p(fb), _p(wb), C.byref(gb), _p(eb))
        assert ga.value == gb.value
        assert ol.same_bits(ea, eb)                                # BSDF_f for the direct-light estimate
        assert (pa == 0.0) == (pb == 0.0), (pa, pb, wo, smp)
        if pb == 0.0:
            zero_pdf += 1
            continue
        tol = 2e-5                                                # a last-ulp change of exp's argument path, amplified by 1 / alpha^2
        assert abs(pa - pb) <= tol * abs(pb), (pa, pb)
        assert np.abs(fa - fb).max() <= tol * max(np.abs(fb).max(), 1e-30), (fa, fb)
        assert np.abs(wa - wb).max() <= 2e-6, (wa, wb)
        inexact += not (np.float32(pa) == np.float32(pb) and ol.same_bits(fa, fb) and ol.same_bits(wa, wb))
        if kind == "GLASS":
            wl = basis.reshape(3, 3) @ wa
            ol_ = basis.reshape(3, 3) @ wo
            refl += (wl[2] * ol_[2]) > 0
            refr += (wl[2] * ol_[2]) < 0
    print(kind, "results not bit-identical to the reference: %d of %d (zero pdf: %d)" % (inexact, n - zero_pdf, zero_pdf))
    assert inexact < 0.05 * n
    assert zero_pdf < 0.6 * n
    if kind == "GLASS":
        assert refl > 50 and refr > 500
