"""How much a spatial-split builder could take off the walk: an upper bound measured by splitting the GEOMETRY.
Triangles whose longest edge is long for the scene are cut at that edge's midpoint (once or twice), the scene is built
with the same SAH builder and the same rays are traced: node visits, triangle tests and kernel time against the uncut scene.
Reference splitting (each triangle kept whole, listed in several leaves with clipped boxes) cannot do better than this:
its clipped boxes are never tighter than the boxes of real pieces.  Hit records of the cut scenes are NOT the scene's
(timing and counters only).
    python tools/exp_presplit.py [--workload courtyard-10M] [--rays 16777216]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np            # noqa: E402
import torch                  # noqa: E402
import bench                  # noqa: E402
from craytracer_amd import api, scenes   # noqa: E402


def split_longest(tri, factor):
    """tri [n,9] -> triangles whose longest edge exceeds factor * the median longest edge cut in two at its midpoint."""
    v = tri.reshape(-1, 3, 3)
    e = np.stack([np.linalg.norm(v[:, 1] - v[:, 0], axis=1), np.linalg.norm(v[:, 2] - v[:, 1], axis=1),
                  np.linalg.norm(v[:, 0] - v[:, 2], axis=1)], 1)
    longest = e.max(1)
    which = e.argmax(1)
    cut = longest > factor * np.median(longest)
    keep = v[~cut]
    c = v[cut]
    w = which[cut]
    idx = np.arange(len(c))
    a = c[idx, w]
    b = c[idx, (w + 1) % 3]
    o = c[idx, (w + 2) % 3]
    m = ((a + b) * np.float32(0.5)).astype(np.float32)
    t1 = np.stack([a, m, o], 1)
    t2 = np.stack([m, b, o], 1)
    return np.concatenate([keep, t1, t2], 0).reshape(-1, 9).astype(np.float32), int(cut.sum())


def measure(si, d_rays, n, device, reps=4):
    built = api.BuiltScene(si, api.ACCEL_BVH4, sah=True)
    sc = api.Scene(built, 0)
    d_hits = torch.empty((n, 5), dtype=torch.int32, device=device)
    st = torch.cuda.current_stream().cuda_stream
    sc.enable_counters(True)
    sc.reset_counters()
    sc.intersect_dev(d_rays.data_ptr(), n, d_hits.data_ptr(), st)
    torch.cuda.synchronize()
    c = sc.counters()
    sc.enable_counters(False)
    ms = []
    for _ in range(reps):
        sc.intersect_dev(d_rays.data_ptr(), n, d_hits.data_ptr(), st)
        torch.cuda.synchronize()
        ms.append(sc.last_kernel_ms())
    sc.close()
    built.close()
    return c, min(ms)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="courtyard-10M")
    ap.add_argument("--rays", type=int, default=1 << 24)
    a = ap.parse_args()
    device = torch.device("cuda", 0)
    si = scenes.courtyard(*bench.WORKLOADS[a.workload], seed=5)
    built = api.BuiltScene(si, api.ACCEL_BVH4, sah=True)
    sc = api.Scene(built, 0)
    d_rays = bench.gen_incoherent_rays(sc, si, a.rays, 0xC0FFEE, device)
    sc.close()
    built.close()
    tri = si.tri_verts
    variants = [("uncut", tri)]
    for factor, levels in ((1.5, 1), (1.2, 1), (1.2, 2), (1.0, 2)):
        t = tri
        cuts = 0
        for _ in range(levels):
            t, k = split_longest(t, factor)
            cuts += k
        variants.append(("factor %.1f x%d (%d cuts)" % (factor, levels, cuts), t))
    for name, t in variants:
        c, ms = measure(api.SceneInput(t), d_rays, a.rays, device)
        print("%-32s triangles %9d  counters %s  walk kernel %.3f ms" % (name, len(t), c, ms), flush=True)


if __name__ == "__main__":
    main()
