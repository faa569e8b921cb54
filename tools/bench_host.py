#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-buffer entry point crt_intersect_batch on the bench workload:
rays start and hits end in host memory (pinned via crt_host_alloc, and ordinary pageable)."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402
from craytracer_amd import api, scenes  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith("--") else "courtyard-10M"
n = 1 << 24
si = scenes.courtyard(*bench.WORKLOADS[wl], seed=5)
sc = api.Scene(api.BuiltScene(si, api.ACCEL_BVH4, sah="--reference-tree" not in sys.argv), 0)
d_rays = bench.gen_incoherent_rays(sc, si, n, 0xC0FFEE, torch.device("cuda", 0))
rays = d_rays.cpu().numpy()
res = {"workload": wl, "rays": n}
hb_r = api.HostBuffer(rays.shape, np.float32)
hb_h = api.HostBuffer(n, api.HIT_DTYPE)
hb_r.array[:] = rays
pageable_hits = np.empty(n, api.HIT_DTYPE)
for name, r, h in (("pinned", hb_r.array, hb_h.array), ("pageable", rays, pageable_hits)):
    sc.intersect(r, out=h)
    ts = []
    for _ in range(5):
        t0 = time.perf_counter()
        sc.intersect(r, out=h)
        ts.append(time.perf_counter() - t0)
    res[name + "_ms"] = round(min(ts) * 1e3, 2)
    res[name + "_mrays"] = round(n / min(ts) / 1e6, 1)
    res[name + "_gbs"] = round(n * 44 / min(ts) / 1e9, 1)
print(json.dumps(res))
