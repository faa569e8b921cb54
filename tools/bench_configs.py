#!/usr/bin/env python3
"""Closest-hit / any-hit rates of the small BASELINE configurations on device-resident rays:
C1 cornell-29 (grid, 512 x 512 primary rays x 16) and C2 analytic shapes (BVH and QBVH, reference
control flow, 1920 x 1080 primary rays x 4)."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from craytracer_amd import api, scenes  # noqa: E402

out = []
cases = [
    ("C1 cornell-29 grid", scenes.cornell29(), api.ACCEL_GRID, scenes.camera_rays(512, 512, (278, 273, 800), (278, 273, 0)), 16),
    ("C2 analytic BVH", scenes.analytic_scene(), api.ACCEL_BVH, scenes.camera_rays(1920, 1080, (0.0, 2.5, 9.0), (0.0, 0.5, 0.0), fov_deg=50.0), 4),
    ("C2 analytic QBVH", scenes.analytic_scene(), api.ACCEL_BVH4, scenes.camera_rays(1920, 1080, (0.0, 2.5, 9.0), (0.0, 0.5, 0.0), fov_deg=50.0), 4),
]
for name, si, accel, rays, rep in cases:
    sc = api.Scene(api.BuiltScene(si, accel), 0)
    r = np.tile(rays, (rep, 1))
    n = len(r)
    d_rays = torch.from_numpy(r).cuda()
    d_hits = torch.empty((n, 5), dtype=torch.int32, device="cuda")
    d_ld = torch.full((n,), 5.0 if "analytic" in name else 300.0, dtype=torch.float32, device="cuda")
    d_t = torch.empty(n, dtype=torch.float32, device="cuda")
    d_occ = torch.empty(n, dtype=torch.uint8, device="cuda")
    ms_c, ms_s = [], []
    for _ in range(4):
        sc.intersect_dev(d_rays.data_ptr(), n, d_hits.data_ptr(), None)
        ms_c.append(sc.last_kernel_ms())
        sc.shadow_dev(d_rays.data_ptr(), d_ld.data_ptr(), n, d_t.data_ptr(), d_occ.data_ptr(), None)
        ms_s.append(sc.last_kernel_ms())
    out.append({"config": name, "rays": n, "mode": sc.traverse_mode(), "closest_ms": round(min(ms_c), 3),
                "closest_mrays": round(n / min(ms_c) / 1e3, 1), "anyhit_ms": round(min(ms_s), 3),
                "anyhit_mrays": round(n / min(ms_s) / 1e3, 1)})
    sc.close()
for o in out:
    print(json.dumps(o))
