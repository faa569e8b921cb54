"""Consecutive, independent launches enqueued in turn on S streams (a library built with -DCRT_SLOTS=S keeps S sets of launch
scratch): ms per step of the full headline batch and of its slowest cell-range share (share 0 of 8) for S = 1 .. 4.
    CRT_GPU_LIB=/path/libcrt_gpu_slots4.so python tools/exp_streams.py [--max-streams 4]"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch                  # noqa: E402
import bench                  # noqa: E402
from craytracer_amd import api, scenes   # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="courtyard-10M")
    ap.add_argument("--rays", type=int, default=1 << 24)
    ap.add_argument("--max-streams", type=int, default=4)
    ap.add_argument("--steps", type=int, default=24)
    a = ap.parse_args()
    device = torch.device("cuda", 0)
    n = a.rays
    si = scenes.courtyard(*bench.WORKLOADS[a.workload], seed=5)
    built = api.BuiltScene(si, api.ACCEL_BVH4, sah=True)
    sc = api.Scene(built, 0)
    d_rays = bench.gen_incoherent_rays(sc, si, n, 0xC0FFEE, device)
    streams = [torch.cuda.Stream(device=device) for _ in range(a.max_streams)]
    hits = [torch.empty((n, 5), dtype=torch.int32, device=device) for _ in range(a.max_streams)]

    def run(S, fn):
        for i in range(2 * S):
            fn(i % S)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(a.steps):
            fn(i % S)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / a.steps * 1e3

    for S in range(1, a.max_streams + 1):
        full = run(S, lambda k: sc.intersect_dev(d_rays.data_ptr(), n, hits[k].data_ptr(), streams[k].cuda_stream))
        share = run(S, lambda k: sc.intersect_dev_slice(d_rays.data_ptr(), n, hits[k].data_ptr(), 0, 8, streams[k].cuda_stream))
        print("streams %d  full batch %.3f ms  share 0 of 8 %.3f ms  ratio %.2f" % (S, full, share, full / share), flush=True)
    same = all(bool((hits[k] == hits[0]).all().item()) for k in range(1, a.max_streams))
    print("result buffers identical:", same)
    sc.close()
    built.close()


if __name__ == "__main__":
    main()
