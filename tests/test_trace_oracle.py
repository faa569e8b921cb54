"""The CPU restatement of the path tracer (oracle/crt_oracle_trace.cpp; PARITY UNPINNED for the control
flow of trace.h, which cannot be compiled here) checked against what can be known without the
reference: the random-number definition, determinism, sample additivity, emission-only images, and
one radiometric closed form (a small rectangle light over a Lambertian floor)."""
import math

import numpy as np

from craytracer_amd import api, scenes
import oraclelib as ol


def fmix32(h):
    h &= 0xFFFFFFFF
    h ^= h >> 16
    h = (h * 0x85EBCA6B) & 0xFFFFFFFF
    h ^= h >> 13
    h = (h * 0xC2B2AE35) & 0xFFFFFFFF
    h ^= h >> 16
    return h


def crt_rng(seed, pixel, sample, dim):
    """DESIGN.md "N1": the definition both the library and the oracle implement."""
    h = fmix32(pixel ^ (seed & 0xFFFFFFFF))
    h = fmix32((h + sample * 0x9E3779B1 + (seed >> 32)) & 0xFFFFFFFF)
    h = fmix32(h ^ ((dim * 0x85EBCA77 + 0x165667B1) & 0xFFFFFFFF))
    return np.float32(h >> 8) * np.float32(1.0 / 16777216.0)


def test_rng_definition():
    L = ol.oracle()
    rs = np.random.default_rng(1)
    for _ in range(2000):
        seed = int(rs.integers(0, 2**63)) * 2 + int(rs.integers(0, 2))
        pixel, sample, dim = (int(x) for x in rs.integers(0, 2**32, 3))
        assert np.float32(L.oracle_rng(seed, pixel, sample, dim)) == crt_rng(seed, pixel, sample, dim)
    u = np.array([L.oracle_rng(5, p, s, d) for p in range(40) for s in range(10) for d in range(12)], np.float64)
    assert u.min() >= 0.0 and u.max() < 1.0
    assert abs(u.mean() - 0.5) < 0.01 and abs(u.var() - 1.0 / 12.0) < 0.005
    uu = u.reshape(40, 10, 12)
    assert abs(np.corrcoef(uu[:, :, 0].ravel(), uu[:, :, 1].ravel())[0, 1]) < 0.1
    assert abs(np.corrcoef(uu[:, 0, :].ravel(), uu[:, 1, :].ravel())[0, 1]) < 0.1


def test_threads_and_sample_ranges():
    si, desc = scenes.cornell29_render(max_depth=4, seed=3)
    cam = ol.oracle_camera(*scenes.CORNELL_CAMERA)
    flm = ol.film("oracle", cam, 0.6911, 40, 30)
    osc = ol.OracleScene(si, api.ACCEL_NONE)
    a = osc.render(desc, cam, flm, 0, 6, nthreads=1)
    b = osc.render(desc, cam, flm, 0, 6, nthreads=8)
    assert ol.same_bits(a, b)
    c = osc.render(desc, cam, flm, 0, 2, nthreads=4)
    c = osc.render(desc, cam, flm, 2, 4, color_buffer=c, nthreads=4)
    assert ol.same_bits(a, c)
    assert np.isfinite(a).all() and a.min() >= 0.0


def test_depth_zero_is_emission_only():
    si, desc = scenes.cornell29_render(max_depth=0, seed=1)
    cam = ol.oracle_camera(*scenes.CORNELL_CAMERA)
    flm = ol.film("oracle", cam, 0.6911, 64, 64)
    img = ol.OracleScene(si, api.ACCEL_NONE).render(desc, cam, flm, 0, 2)
    assert set(np.unique(img[..., 0]).tolist()) <= {0.0, 90.0, 180.0}
    lit = img[..., 0] == 180.0
    assert lit.sum() > 10
    assert np.allclose(img[lit], 2 * 90.0 * np.array([1.0, 0.85, 0.5], np.float32))


def test_small_light_over_lambertian_floor():
    """Direct light of a small square emitter of radiance Le at height h over a Lambertian floor,
    seen at the point below its centre: L = (cd / pi) * Le * A / h^2 to first order in A / h^2
    (both cosines ~ 1).  max_depth 1, one light, so good_paths == 1 and the estimator's mean is L."""
    sl = scenes.ShapeList()
    sl.rectangle((-50.0, 0.0, 50.0), (100.0, 0.0, 0.0), (0.0, 0.0, -100.0))                   # floor, normal +y
    a, h = 0.2, 4.0
    sl.rectangle((a / 2, h, a / 2), (-a, 0.0, 0.0), (0.0, 0.0, -a))                            # light, normal -y
    si = sl.to_input()
    cd, le = 0.6, 50.0
    mats = np.array([api.material(api.MAT_MATTE, (cd, cd, cd)), api.material(api.MAT_EMISSIVE, (1.0, 1.0, 1.0), le)],
                    api.MATERIAL_DTYPE)
    desc = api.RenderDesc(si, [0, 1], mats, 1, seed=2)
    cam = ol.oracle_camera((3.0, 2.0, 0.0), (0.0, 0.0, 0.0), focal_pt_dist=1.0)
    flm = ol.film("oracle", cam, 0.002, 4, 4)                         # 16 pixels, all looking at the origin
    n = 4000
    img = ol.OracleScene(si, api.ACCEL_NONE).render(desc, cam, flm, 0, n) / n
    want = cd / math.pi * le * a * a / (h * h)
    assert abs(img.mean() - want) / want < 0.02
