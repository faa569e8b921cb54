"""Hit records of the same rays through the reference's median-split QBVH and through the SAH-built one (CRT_BUILD_SAH):
how many rays differ, and whether every difference is an exact-t tie (two objects hit at bit-identical t).
    python tools/compare_trees.py [--workload courtyard-10M] [--rays 16777216]"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np            # noqa: E402
import torch                  # noqa: E402
import bench                  # noqa: E402
from craytracer_amd import api, scenes   # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="courtyard-10M")
    ap.add_argument("--rays", type=int, default=1 << 24)
    a = ap.parse_args()
    device = torch.device("cuda", 0)
    si = scenes.courtyard(*bench.WORKLOADS[a.workload], seed=5)
    res = {}
    d_rays = None
    for name, sah in (("reference", False), ("sah", True)):
        built = api.BuiltScene(si, api.ACCEL_BVH4, sah=sah)
        sc = api.Scene(built, 0)
        if d_rays is None:
            d_rays = bench.gen_incoherent_rays(sc, si, a.rays, 0xC0FFEE, device)
        d_hits = torch.empty((a.rays, 5), dtype=torch.int32, device=device)
        sc.intersect_dev(d_rays.data_ptr(), a.rays, d_hits.data_ptr(), torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        res[name] = d_hits.cpu().numpy().view(api.HIT_DTYPE).reshape(-1)
        sc.close()
        built.close()
    r, s = res["reference"], res["sah"]
    same = (r["t"].view(np.uint32) == s["t"].view(np.uint32)) & (r["obj"] == s["obj"]) & (r["beta"].view(np.uint32) == s["beta"].view(np.uint32)) & \
        (r["gamma"].view(np.uint32) == s["gamma"].view(np.uint32))
    diff = ~same
    tie = diff & (r["t"].view(np.uint32) == s["t"].view(np.uint32)) & (r["obj"] != s["obj"])
    print(json.dumps({"rays": int(a.rays), "differing": int(diff.sum()), "exact_t_ties": int(tie.sum()),
                      "other": int((diff & ~tie).sum()), "hit_fraction": float((r["obj"] >= 0).mean())}))


if __name__ == "__main__":
    main()
