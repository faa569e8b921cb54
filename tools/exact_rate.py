import sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, bench
from craytracer_amd import api, scenes
si = scenes.courtyard(64, 64, 128, seed=5)
sc = api.Scene(api.BuiltScene(si, api.ACCEL_BVH4), 0)
n = 1 << 22
d_rays = bench.gen_incoherent_rays(sc, si, n, 1, torch.device("cuda", 0))
d_hits = torch.empty((n, 5), dtype=torch.int32, device="cuda")
for mode in (api.TRAVERSE_PRUNED, api.TRAVERSE_EXACT):
    sc.set_traverse_mode(mode)
    sc.enable_counters(True); sc.reset_counters()
    sc.intersect_dev(d_rays.data_ptr(), n, d_hits.data_ptr(), None)
    c = sc.counters(); sc.enable_counters(False)
    ms = []
    for _ in range(3):
        sc.intersect_dev(d_rays.data_ptr(), n, d_hits.data_ptr(), None)
        ms.append(sc.last_kernel_ms())
    print("mode", mode, "ms", min(ms), "Mrays/s", n / min(ms) / 1e3, "nodes/ray", c["nodes_visited"] / n, "prims/ray", c["prims_tested"] / n)
