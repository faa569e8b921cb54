"""samples/sec/pixel of the wavefront path tracer (crt_renderer_*) on the courtyard stand-in, QBVH.
    python tools/bench_render.py [--copies 640 --lat 64 --lon 128] [--width 3840 --height 2160] [--spp 4] [--depth 5]
Prints one JSON line.  (The comparison with the CPU oracle lives in tests/test_gpu_render.py and in bench.py's cpu_baseline leg.)"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np                                    # noqa: E402
from craytracer_amd import api, scenes                # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--copies", type=int, default=640)       # bench.py's courtyard-10M
    ap.add_argument("--lat", type=int, default=64)
    ap.add_argument("--lon", type=int, default=128)
    ap.add_argument("--width", type=int, default=3840)
    ap.add_argument("--height", type=int, default=2160)
    ap.add_argument("--spp", type=int, default=4)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--depth", type=int, default=5)
    ap.add_argument("--in-flight", type=int, default=0)
    ap.add_argument("--glossy", action="store_true", help="blobs in PLASTIC / METAL / GLASS instead of MATTE")
    ap.add_argument("--builder", default="sah", choices=["sah", "reference"])
    a = ap.parse_args()
    t0 = time.time()
    si, desc, pos, look = scenes.courtyard_render(a.copies, a.lat, a.lon, max_depth=a.depth, samples_in_flight=a.in_flight, glossy=a.glossy)
    built = api.BuiltScene(si, api.ACCEL_BVH4, sah=a.builder == "sah")
    scene = api.Scene(built)
    t_build = time.time() - t0
    cam = api.make_camera(pos, look, focal_pt_dist=0.05)
    flm = api.make_film(cam, 1.0, a.width, a.height)
    r = api.Renderer(scene, desc, cam, flm)
    if a.warmup:
        r.render(a.warmup)
    s0 = r.stats()
    t1 = time.time()
    r.render(a.spp)
    dt = time.time() - t1
    s1 = r.stats()
    img, ns = r.read()
    rays = (s1["rays_closest"] - s0["rays_closest"]) + (s1["rays_shadow"] - s0["rays_shadow"])
    out = {
        "metric": "samples/sec/pixel", "value": a.spp / dt, "workload": "courtyard-%dk-open, %s, QBVH (%s split), %dx%d, depth %d" % (
            si.num_obj // 1000, "plastic / metal / glass blobs" if a.glossy else "matte", a.builder, a.width, a.height, a.depth),
        "spp_timed": a.spp, "seconds": dt, "mrays_per_s_all_queries": rays / dt / 1e6,
        "rays_closest": s1["rays_closest"] - s0["rays_closest"], "rays_shadow": s1["rays_shadow"] - s0["rays_shadow"],
        "ms_extend_kernels": s1["ms_extend"] - s0["ms_extend"], "ms_shadow_kernels": s1["ms_shadow"] - s0["ms_shadow"],
        "ms_total": s1["ms_total"] - s0["ms_total"], "build_seconds": t_build,
        "mean_radiance": float(img.mean() / ns), "finite": bool(np.isfinite(img).all()),
    }
    print(json.dumps(out))


if __name__ == "__main__":
    main()
