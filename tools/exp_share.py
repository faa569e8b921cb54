"""Experiment: what a 2^21-ray share of the headline batch costs, and why (r03).  Walk / binning / wall time of one share of 8
under knobs: streams 1 / 2 / 4, binning on / off (index slices), tail splitting on / off.  Usage: python tools/exp_share.py [--lib path]"""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser()
ap.add_argument("--lib", default=None)
ap.add_argument("--rays", type=int, default=1 << 24)
ap.add_argument("--slices", type=int, default=8)
args = ap.parse_args()
if args.lib:
    os.environ["CRT_GPU_LIB"] = args.lib
import numpy as np
import torch
import bench
from craytracer_amd import api, scenes
dev = torch.device("cuda", 0)
si = scenes.courtyard(*bench.WORKLOADS["courtyard-10M"], seed=5)
built = api.BuiltScene(si, api.ACCEL_BVH4, sah=True)
sc = api.Scene(built, 0)
n, K = args.rays, args.slices
rays = bench.gen_incoherent_rays(sc, si, n, 0xC0FFEE, dev)
hits = [torch.empty((n, 5), dtype=torch.int32, device=dev) for _ in range(4)]
streams = [torch.cuda.Stream() for _ in range(4)]
res = {}
def run(name, fn, ns, steps=16):
    for i in range(4):
        fn(streams[i % ns].cuda_stream, i % ns)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        fn(streams[i % ns].cuda_stream, i % ns)
    torch.cuda.synchronize()
    res[name] = round((time.perf_counter() - t0) / steps * 1e3, 4)
    print(name, res[name], flush=True)
for ns in (1, 2, 4):
    run("full/%d" % ns, lambda st, w: sc.intersect_dev(rays.data_ptr(), n, hits[w].data_ptr(), st), ns, 8)
    for k in (0, 5):
        run("share%d/%d" % (k, ns), lambda st, w: sc.intersect_dev_slice(rays.data_ptr(), n, hits[w].data_ptr(), k, K, st), ns)
    m = n // K
    run("own-batch-of-%d/%d" % (m, ns), lambda st, w: sc.intersect_dev(rays.data_ptr(), m, hits[w].data_ptr(), st), ns)
sc.set_ray_binning(False)
for ns in (1, 2):
    run("index-share-unbinned/%d" % ns, lambda st, w: sc.intersect_dev_slice(rays.data_ptr(), n, hits[w].data_ptr(), 3, K, st), ns)
sc.set_ray_binning(True)
print(json.dumps(res))
