"""Experiment (r03): the 8-wide walk against the 4-wide one on the headline workload -- same hit records?  how fast?
Usage: python tools/exp_wide8.py [--lib path] [--rays N] [--scene courtyard-10M|courtyard-1M]"""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser()
ap.add_argument("--lib", default=None)
ap.add_argument("--rays", type=int, default=1 << 24)
ap.add_argument("--scene", default="courtyard-10M")
ap.add_argument("--builder", default="sah")
ap.add_argument("--small", type=int, default=1 << 21)
args = ap.parse_args()
os.environ["CRT_WIDE8"] = "1"
if args.lib:
    os.environ["CRT_GPU_LIB"] = args.lib
import numpy as np
import torch
import bench
from craytracer_amd import api, scenes
dev = torch.device("cuda", 0)
si = scenes.courtyard(*bench.WORKLOADS[args.scene], seed=5)
t0 = time.time()
built = api.BuiltScene(si, api.ACCEL_BVH4, sah=(args.builder == "sah"))
sc = api.Scene(built, 0)
print("scene built and uploaded in %.1f s" % (time.time() - t0), flush=True)
n = args.rays
sc.set_wide_nodes(False)
rays = bench.gen_incoherent_rays(sc, si, n, 0xC0FFEE, dev)
res = {}
hits = {}
for wide in (False, True, False, True):
    sc.set_wide_nodes(wide)
    h = torch.empty((n, 5), dtype=torch.int32, device=dev)
    for m in (n, args.small):
        for _ in range(2):
            sc.intersect_dev(rays.data_ptr(), m, h.data_ptr(), 0)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        K = 8
        for _ in range(K):
            sc.intersect_dev(rays.data_ptr(), m, h.data_ptr(), 0)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / K * 1e3
        key = "%s/%d" % ("wide8" if wide else "wide4", m)
        res.setdefault(key, []).append(round(ms, 4))
        print(key, round(ms, 4), "ms  kernel", round(sc.last_kernel_ms(), 4), " %.1f Mrays/s" % (m / ms / 1e3), flush=True)
    sc.intersect_dev(rays.data_ptr(), n, h.data_ptr(), 0)
    torch.cuda.synchronize()
    hits[wide] = h
same = bool(torch.equal(hits[False], hits[True]))
print("hit records identical:", same, flush=True)
if not same:
    d = (hits[False] != hits[True]).any(dim=1)
    idx = torch.nonzero(d).flatten()
    print("differing records:", int(d.sum()), idx[:8].tolist())
    print(hits[False][idx[:8]].cpu().numpy().view(np.float32)[:, 0], hits[True][idx[:8]].cpu().numpy().view(np.float32)[:, 0])
    print(hits[False][idx[:8]].cpu().numpy(), hits[True][idx[:8]].cpu().numpy())
for wide in (False, True):
    sc.set_wide_nodes(wide)
    sc.enable_counters(True); sc.reset_counters()
    sc.intersect_dev(rays.data_ptr(), n, hits[wide].data_ptr(), 0)
    torch.cuda.synchronize()
    c = sc.counters()
    print("wide8" if wide else "wide4", "per ray: nodes %.3f prims %.3f (rays %d)" % (c["nodes_visited"] / n, c["prims_tested"] / n, c["rays_closest"]), flush=True)
    sc.enable_counters(False)
print(json.dumps(res))
