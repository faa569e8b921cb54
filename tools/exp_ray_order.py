"""Experiment: how much would a finer ray order than the library's 16^3-cell counting sort buy the closest-hit walk?
Rays are sorted here (torch, on the device) by a Morton key of their origin at 2^bits cells per axis, optionally with the
direction octant as the lowest key bits, and traced with the library's own binning OFF; kernel time by HIP events.
usage: python3 tools/exp_ray_order.py [workload] [builder]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import bench
from craytracer_amd import api, scenes

workload = sys.argv[1] if len(sys.argv) > 1 else "courtyard-10M"
sah = (sys.argv[2] if len(sys.argv) > 2 else "sah") == "sah"
dev = torch.device("cuda", 0)
si = scenes.courtyard(*bench.WORKLOADS[workload], seed=5)
sc = api.Scene(api.BuiltScene(si, api.ACCEL_BVH4, sah=sah), 0)
n = 1 << 24
rays = bench.gen_incoherent_rays(sc, si, n, 0xC0FFEE, dev)
hits = torch.empty((n, 5), dtype=torch.int32, device=dev)
stream = torch.cuda.current_stream().cuda_stream
lo = torch.tensor(si.tri_verts.reshape(-1, 3).min(axis=0), device=dev)
hi = torch.tensor(si.tri_verts.reshape(-1, 3).max(axis=0), device=dev)


def spread(v, bits):
    out = torch.zeros_like(v)
    for b in range(bits):
        out |= ((v >> b) & 1) << (3 * b)
    return out


def ordered(bits, octant, morton=True):
    c = ((rays[:, 0:3] - lo) / (hi - lo) * (1 << bits)).floor().clamp(0, (1 << bits) - 1).to(torch.int64)
    if morton:
        key = spread(c[:, 0], bits) | (spread(c[:, 1], bits) << 1) | (spread(c[:, 2], bits) << 2)
    else:
        key = (c[:, 2] << (2 * bits)) | (c[:, 1] << bits) | c[:, 0]
    if octant:
        o = (rays[:, 3] > 0).to(torch.int64) | ((rays[:, 4] > 0).to(torch.int64) << 1) | ((rays[:, 5] > 0).to(torch.int64) << 2)
        key = (key << 3) | o
    return rays[torch.argsort(key, stable=True)].contiguous()


def run(r, binning, label):
    sc.set_ray_binning(binning)
    ms = []
    for _ in range(5):
        sc.intersect_dev(r.data_ptr(), n, hits.data_ptr(), stream)
        ms.append(sc.last_kernel_ms())
    torch.cuda.synchronize()
    chk = int(hits.to(torch.int64).sum().item()) & 0xFFFFFFFFFFFF
    print("%-44s kernel %.3f ms  (%.0f Mrays/s kernel only)  chk %d" % (label, float(np.mean(ms[1:])), n / float(np.mean(ms[1:])) / 1e3, chk), flush=True)


run(rays, True, "library binning (16^3 cells, counting sort)")
run(rays, False, "as generated, no binning")
m4 = ordered(4, False)
run(m4, False, "morton 4 bits/axis, physically sorted (stable)")
run(m4, True, "  the same rays through library binning")
c = ((rays[:, 0:3] - lo) / (hi - lo) * 16).floor().clamp(0, 15).to(torch.int64)
key = spread(c[:, 0], 4) | (spread(c[:, 1], 4) << 1) | (spread(c[:, 2], 4) << 2)
shuf = torch.randperm(n, device=dev)
m4r = rays[shuf][torch.argsort(key[shuf], stable=True)].contiguous()
run(m4r, False, "morton 4 bits/axis, random order inside a cell")
run(ordered(8, False), False, "morton 8 bits/axis, physically sorted")
