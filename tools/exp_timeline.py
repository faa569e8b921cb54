"""Experiment (r03): where a small launch's time goes.  With the diagnostic build (make variant VARIANT=-DCRT_DIAG=1) each wave of
the headline walk notes the device clock when it starts, when its queue runs dry and when it ends; crt_scene_get_counters prints
the launch's timeline.  Usage: python tools/exp_timeline.py --lib variants/diag.so [--split 0|1]"""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser()
ap.add_argument("--lib", default=None)
ap.add_argument("--split", type=int, default=1)
args = ap.parse_args()
if args.lib:
    os.environ["CRT_GPU_LIB"] = args.lib
if not args.split:
    os.environ["CRT_TAIL_SPLIT_MAX"] = "0"
import torch
import bench
from craytracer_amd import api, scenes
dev = torch.device("cuda", 0)
si = scenes.courtyard(*bench.WORKLOADS["courtyard-10M"], seed=5)
built = api.BuiltScene(si, api.ACCEL_BVH4, sah=True)
sc = api.Scene(built, 0)
n = 1 << 24
rays = bench.gen_incoherent_rays(sc, si, n, 0xC0FFEE, dev)
h = torch.empty((n, 5), dtype=torch.int32, device=dev)
sc.enable_counters(True)
for label, fn in (("full batch of 2^24", lambda: sc.intersect_dev(rays.data_ptr(), n, h.data_ptr(), None)),
                  ("share 3 of 8 (cell range)", lambda: sc.intersect_dev_slice(rays.data_ptr(), n, h.data_ptr(), 3, 8, None)),
                  ("own batch of 2^21", lambda: sc.intersect_dev(rays.data_ptr(), n >> 3, h.data_ptr(), None)),
                  ("own batch of 2^20", lambda: sc.intersect_dev(rays.data_ptr(), n >> 4, h.data_ptr(), None))):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    sc.reset_counters()
    fn()
    torch.cuda.synchronize()
    print("==== %s: walk kernel %.4f ms, binning %.4f ms" % (label, sc.last_kernel_ms(), sc.last_binning_ms()), flush=True)
    sys.stderr.flush()
    sc.counters()
    sys.stderr.flush()
