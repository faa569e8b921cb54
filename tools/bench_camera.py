#!/usr/bin/env python3
"""crt_primary_rays_dev timing: 3840 x 2160 pixels, pinhole and thin lens, with per-pixel samples."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from craytracer_amd import api  # noqa: E402

w, h = 3840, 2160
n = w * h
d_rays = torch.empty((n, 6), dtype=torch.float32, device="cuda")
d_is = torch.rand((n, 2), device="cuda")
d_ls = torch.rand((n, 2), device="cuda")
for name, lens in (("pinhole", None), ("thin_lens", (8.0, 0.12))):
    cam = api.make_camera((0.0, 2.5, 9.0), (0.0, 0.5, 0.0), (0.0, 1.0, 0.0), 0.05, lens)
    flm = api.make_film(cam, 0.9, w, h)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    st = torch.cuda.current_stream()
    for _ in range(3):
        api.primary_rays_dev(cam, flm, 0, n, d_rays.data_ptr(), d_is.data_ptr(), d_ls.data_ptr(), stream=st.cuda_stream)
    ev[0].record(st)
    for _ in range(20):
        api.primary_rays_dev(cam, flm, 0, n, d_rays.data_ptr(), d_is.data_ptr(), d_ls.data_ptr(), stream=st.cuda_stream)
    ev[1].record(st)
    torch.cuda.synchronize()
    ms = ev[0].elapsed_time(ev[1]) / 20
    print(json.dumps({"camera": name, "pixels": n, "ms": round(ms, 4), "grays_per_s": round(n / ms / 1e6, 2),
                      "algorithmic_GBs": round(n * (24 + 16) / ms / 1e6, 1)}))
