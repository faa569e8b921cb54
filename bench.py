#!/usr/bin/env python3
"""bench.py -- incoherent closest-hit throughput of the QBVH traversal kernel on MI355X.

  python bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path (n x intersectTest, /root/reference/intersect.h:365, BVH4
branch) over one batch of 2^24 incoherent rays that is already resident in HBM.  The workload is
BASELINE.json's headline configuration (configs[4]: "San Miguel ~10M tris, QBVH, incoherent diffuse
bounces") on the procedural stand-in scenes.courtyard (the real asset is not shipped and cannot be
fetched): 10 485 772 world-space triangles, QBVH built with the reference's rules, ray set
R-incoherent = shuffled bounce-1..3 cosine-hemisphere rays.

Multi-GPU: one process per GPU, full scene replica per GPU, each rank traces its own 2^24-ray batch
(weak scaling, no data-path collective -- rays are independent); timing = max over ranks.

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` and `cpu_baseline`.
Nothing here reads /root/reference.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s (spec)

WORKLOADS = {
    # name: (n_copies, n_lat, n_lon) -> 2*n_lat*n_lon*n_copies + 12 triangles
    "courtyard-10M": (640, 64, 128),
    "courtyard-1M": (64, 64, 128),
    "courtyard-64k": (4, 64, 128),
}


def gen_incoherent_rays(sc, si, n_rays, seed, device):
    """R-incoherent on the GPU engine itself: random primary rays inside the scene bounds, then
    cosine-hemisphere bounces from the hit points (origin = hit point, no offset), bounce 1..3,
    shuffled.  Ray generation is input synthesis, outside the timed region; it runs with the
    counters on, i.e. through the counting instantiation of the kernel, so that in a rocprofv3
    kernel summary the plain instantiation holds the warm-up and timed launches only."""
    import torch
    from craytracer_amd import api
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    tri = torch.from_numpy(si.tri_verts).to(device).view(-1, 3, 3)
    lo = tri.view(-1, 3).min(0).values
    hi = tri.view(-1, 3).max(0).values
    n0 = int(n_rays / 3 * 1.15) + 1024
    o = torch.rand((n0, 3), generator=g, device=device) * (hi - lo) + lo
    d = torch.randn((n0, 3), generator=g, device=device)
    d = d / d.norm(dim=1, keepdim=True)
    rays = torch.cat([o, d], 1).float().contiguous()
    out = []
    sc.enable_counters(True)
    for _ in range(3):
        hits = torch.empty((rays.shape[0], 5), dtype=torch.int32, device=device)
        sc.intersect_dev(rays.data_ptr(), rays.shape[0], hits.data_ptr(), torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        obj = hits[:, 1].long()
        ok = obj >= 0
        t = hits[:, 0].view(torch.float32)
        r = rays[ok]
        tt = t[ok]
        tv = tri[obj[ok]]
        nrm = torch.cross(tv[:, 1] - tv[:, 0], tv[:, 2] - tv[:, 0], dim=1)
        nrm = nrm / nrm.norm(dim=1, keepdim=True).clamp_min(1e-30)
        flip = (nrm * r[:, 3:]).sum(1) > 0
        nrm[flip] = -nrm[flip]
        p = r[:, :3] + r[:, 3:] * tt[:, None]
        u1 = torch.rand(r.shape[0], generator=g, device=device)
        u2 = torch.rand(r.shape[0], generator=g, device=device)
        rr, ph = u1.sqrt(), 2 * np.pi * u2
        a = torch.where(nrm[:, :1].abs() > 0.9, torch.tensor([[0.0, 1.0, 0.0]], device=device),
                        torch.tensor([[1.0, 0.0, 0.0]], device=device))
        tx = torch.cross(a.expand_as(nrm), nrm, dim=1)
        tx = tx / tx.norm(dim=1, keepdim=True)
        ty = torch.cross(nrm, tx, dim=1)
        dd = tx * (rr * ph.cos())[:, None] + ty * (rr * ph.sin())[:, None] + nrm * (1 - u1).clamp_min(0).sqrt()[:, None]
        dd = dd / dd.norm(dim=1, keepdim=True)
        rays = torch.cat([p, dd], 1).float().contiguous()
        out.append(rays)
    sc.enable_counters(False)
    sc.reset_counters()
    allr = torch.cat(out, 0)
    perm = torch.randperm(allr.shape[0], generator=g, device=device)
    allr = allr[perm]
    if allr.shape[0] < n_rays:                      # top up with fresh random rays (rare: very open scenes)
        extra = n_rays - allr.shape[0]
        o = torch.rand((extra, 3), generator=g, device=device) * (hi - lo) + lo
        d = torch.randn((extra, 3), generator=g, device=device)
        allr = torch.cat([allr, torch.cat([o, d / d.norm(dim=1, keepdim=True)], 1).float()], 0)
    return allr[:n_rays].contiguous()


def gen_shadow_rays(sc, si, d_rays, seed, device):
    """R-shadow: from the hit points of the incoherent rays toward uniformly sampled
    points on a rectangle light under the ceiling; light_dist = distance (trace.h:253-260,:337).
    Origin = hit point, no offset (the reference offsets nothing)."""
    import torch
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    n = d_rays.shape[0]
    tri = torch.from_numpy(si.tri_verts).to(device).view(-1, 3)
    lo, hi = tri.min(0).values, tri.max(0).values
    hits = torch.empty((n, 5), dtype=torch.int32, device=device)
    sc.enable_counters(True)
    sc.intersect_dev(d_rays.data_ptr(), n, hits.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    sc.enable_counters(False)
    sc.reset_counters()
    t = hits[:, 0].view(torch.float32)
    ok = hits[:, 1] >= 0
    p = torch.where(ok[:, None], d_rays[:, :3] + d_rays[:, 3:] * t[:, None], d_rays[:, :3])
    c = (lo + hi) * 0.5
    ext = (hi - lo) * 0.1
    q = torch.empty((n, 3), device=device)
    q[:, 0] = c[0] + (torch.rand(n, generator=g, device=device) * 2 - 1) * ext[0]
    q[:, 2] = c[2] + (torch.rand(n, generator=g, device=device) * 2 - 1) * ext[2]
    q[:, 1] = hi[1] - (hi[1] - lo[1]) * 0.002
    v = q - p
    ld = v.norm(dim=1).clamp_min(1e-6)
    rays = torch.cat([p, v / ld[:, None]], 1).float().contiguous()
    return rays, ld.float().contiguous()


def render_leg(args, api, scenes, shard, torch, rank, world, local_rank, device, barrier, n_copies, n_lat, n_lon):
    import torch.distributed as dist
    width, height, depth, spp = 3840, 2160, 5, args.render_spp
    si, desc, pos, look = scenes.courtyard_render(n_copies, n_lat, n_lon, seed=5, max_depth=depth, rng_seed=1)
    built = api.BuiltScene(si, api.ACCEL_BVH4, sah=(args.builder == "sah"))
    sc = api.Scene(built, local_rank)
    cam = api.make_camera(pos, look, focal_pt_dist=0.05)
    flm = api.make_film(cam, 1.0, width, height)
    r = api.Renderer(sc, desc, cam, flm)
    r.render(1)                                           # warm-up pass
    barrier()
    s0 = r.stats()
    t0 = time.perf_counter()
    # rank k renders samples [k*spp, (k+1)*spp) of the one image; RCCL all-reduce of the colour buffers (world > 1)
    d_img, total_spp = shard.render_sharded(r, spp, rank, world, device)
    barrier()
    dt = shard.max_over_ranks(time.perf_counter() - t0, device)
    s1 = r.stats()
    rays = (s1["rays_closest"] - s0["rays_closest"]) + (s1["rays_shadow"] - s0["rays_shadow"])
    mean = float(d_img.mean().item()) / total_spp
    res = {"metric": "samples/sec/pixel, wavefront path tracing (matte + env light), QBVH", "value": round(spp * world / dt, 3),
           "unit": "samples/s/pixel", "workload": "courtyard opened to a constant sky, %d triangles, %dx%d, max depth %d, %d spp per GPU timed"
           % (si.num_obj, width, height, depth, spp), "builder": args.builder, "seconds": round(dt, 4),
           "mrays_per_s_all_queries_per_gpu": round(rays / (s1["ms_total"] - s0["ms_total"]) / 1e3, 1),
           "traversal_kernel_fraction": round(((s1["ms_extend"] - s0["ms_extend"]) + (s1["ms_shadow"] - s0["ms_shadow"]))
                                              / (s1["ms_total"] - s0["ms_total"]), 3),
           "rays_per_sample_per_pixel": round(rays / spp / (width * height), 3), "mean_radiance": round(mean, 5),
           "reduce": "RCCL all_reduce of the colour buffers" if world > 1 else "none (1 GPU)"}
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import oraclelib as ol
        cores = min(os.cpu_count() or 1, int(os.environ.get("OMP_NUM_THREADS", "16")))
        w2, h2 = 256, 144
        flm2 = api.make_film(cam, 1.0, w2, h2)
        o = ol.OracleScene(si, built=built)
        t1 = time.perf_counter()
        want = o.render(desc, cam, flm2, 0, 2, nthreads=cores)
        cpu_s = time.perf_counter() - t1
        r2 = api.Renderer(sc, desc, cam, flm2)
        r2.render(2)
        got, _ = r2.read()
        r2.close()
        res["cpu_baseline"] = {"value": round(2 * w2 * h2 / cpu_s / (width * height), 5), "unit": "samples/s/pixel at %dx%d" % (width, height),
                               "cores": cores, "kind": "port", "sample": "%dx%d image, 2 spp, oracle_render, %.1f s" % (w2, h2, cpu_s),
                               "gpu_image_bit_identical_on_sample": bool((got == want).all())}
        o.close()
    r.close()
    sc.close()
    built.close()
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="courtyard-10M", choices=sorted(WORKLOADS))
    ap.add_argument("--rays", type=int, default=1 << 24)
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="target wall time of the cpu_baseline sample")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--binning", default="on", choices=["on", "off"],
                    help="the library's ray binning step (inside the timed region); off = arrival order")
    ap.add_argument("--accel", default="bvh4", choices=["bvh4", "bvh", "grid"],
                    help="bvh4 = the headline QBVH path; bvh / grid = the reference's other accelerators (extra lines)")
    ap.add_argument("--builder", default=None, choices=["reference", "sah"],
                    help="sah (default for bvh4) = CRT_BUILD_SAH: same QBVH node format and traversal, split planes by a surface-area "
                         "heuristic; reference = the reference's median-split builder (bvh4.h:127-297), reported beside it by default")
    ap.add_argument("--no-reference-tree", action="store_true",
                    help="skip the leg that repeats the batch on the reference's median-split tree (profiling runs: it launches "
                         "the same kernel, so per-kernel profile averages would mix the two trees)")
    ap.add_argument("--no-render", action="store_true", help="skip the samples/sec/pixel leg (the wavefront path tracer)")
    ap.add_argument("--render-spp", type=int, default=4)
    ap.add_argument("--mode", default="closest", choices=["closest", "shadow"],
                    help="closest = the headline metric; shadow = the secondary any-hit metric")
    args = ap.parse_args()
    if args.builder is None:
        args.builder = "sah" if args.accel == "bvh4" else "reference"
    if args.builder == "sah" and args.accel != "bvh4":
        raise SystemExit("--builder sah is a QBVH option")

    import torch
    from craytracer_amd import api, scenes, shard

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: there is no CPU fallback for the traversal path")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)

    # ---- scene (full replica per GPU) ----
    t0 = time.time()
    n_copies, n_lat, n_lon = WORKLOADS[args.workload]
    si = scenes.courtyard(n_copies, n_lat, n_lon, seed=5)
    accel = {"bvh4": api.ACCEL_BVH4, "bvh": api.ACCEL_BVH, "grid": api.ACCEL_GRID}[args.accel]
    built = api.BuiltScene(si, accel, sah=(args.builder == "sah"))
    sc = api.Scene(built, local_rank)
    sc.set_ray_binning(args.binning == "on")
    build_s = time.time() - t0
    n_tris = si.num_obj

    # ---- rays: this rank's own batch, resident in HBM ----
    n = args.rays
    d_rays = gen_incoherent_rays(sc, si, n, 0xC0FFEE + rank, device)
    d_hits = torch.empty((n, 5), dtype=torch.int32, device=device)
    stream = torch.cuda.current_stream().cuda_stream
    shadow = args.mode == "shadow"
    if shadow:
        d_rays, d_ld = gen_shadow_rays(sc, si, d_rays, 0xBEEF + rank, device)
        d_t = torch.empty(n, dtype=torch.float32, device=device)
        d_occ = torch.empty(n, dtype=torch.uint8, device=device)

    def step():
        if shadow:
            sc.shadow_dev(d_rays.data_ptr(), d_ld.data_ptr(), n, d_t.data_ptr(), d_occ.data_ptr(), stream)
        else:
            sc.intersect_dev(d_rays.data_ptr(), n, d_hits.data_ptr(), stream)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # per-ray traversal statistics (defines the algorithmic bytes), outside the timed region
    sc.enable_counters(True)
    sc.reset_counters()
    step()
    cnt = sc.counters()
    sc.enable_counters(False)
    nodes_per_ray = cnt["nodes_visited"] / n
    tris_per_ray = cnt["prims_tested"] / n
    # 24 (ray in) + 20 (hit out) + one QBVH node per visit + 36 + 4 per triangle tested.
    # A node visit reads 112 of the node's 128 bytes (6 plane rows + the child words; the axis byte
    # travels in the parent's child word), so 112 is what is charged, not SURVEY's planning value 128.
    io_bytes = 28 + 5 if shadow else 24 + 20          # shadow: ray + light_dist in, t + occluded out
    if args.accel == "bvh4":
        bytes_per_ray = io_bytes + 112 * nodes_per_ray + 40 * tris_per_ray
    elif args.accel == "bvh":                          # 32-byte nodes
        bytes_per_ray = io_bytes + 32 * nodes_per_ray + 40 * tris_per_ray
    else:                                              # grid: nodes_visited counts cells walked; 8 B of cell_start
        bytes_per_ray = io_bytes + 8 * nodes_per_ray + 44 * tris_per_ray   # + 4 B item index per object tested

    for _ in range(args.warmup):
        step()
    barrier()
    kernel_ms, bin_ms = [], []
    t_start = time.perf_counter()
    for _ in range(args.steps):
        step()
        kernel_ms.append(sc.last_kernel_ms())      # HIP events on the launch stream (syncs the step)
        bin_ms.append(sc.last_binning_ms())
    barrier()
    elapsed = time.perf_counter() - t_start
    elapsed = shard.max_over_ranks(elapsed, device)
    if shadow:
        hit_frac = float(d_occ.float().mean().item())
        checksum = int(d_t.view(torch.int32).to(torch.int64).sum().item() + d_occ.to(torch.int64).sum().item()) & 0xFFFFFFFFFFFF
    else:
        hit_frac = float((d_hits[:, 1] >= 0).float().mean().item())
        checksum = int(d_hits.to(torch.int64).sum().item()) & 0xFFFFFFFFFFFF  # same scene + rays => same number

    gpu_hits_host = None if shadow else d_hits.cpu().numpy().view(api.HIT_DTYPE).reshape(-1)
    host_rays_keep = d_rays.cpu().numpy()
    total_rays = float(n) * world * args.steps
    value = total_rays / elapsed / 1e6
    avg_kernel_ms = float(np.mean(kernel_ms))
    achieved_gbs = bytes_per_ray * n / (avg_kernel_ms * 1e-3) / 1e9

    out = {
        "metric": ("Mrays/s any-hit (shadow rays toward an area light), %s" if shadow else "Mrays/s closest-hit (incoherent), %s")
        % {"bvh4": "QBVH", "bvh": "binary BVH", "grid": "uniform grid"}[args.accel],
        "value": round(value, 3), "unit": "Mrays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "%s: %d-triangle procedural courtyard (San Miguel stand-in), %s, %d %s per GPU per step"
                               % (args.workload, n_tris, {"bvh4": "QBVH (%s)" % ("SAH split planes" if args.builder == "sah" else "the reference's median split"),
                                                            "bvh": "binary BVH", "grid": "uniform grid"}[args.accel], n, "shadow rays from the incoherent hit points toward a ceiling light"
                                  if shadow else "incoherent bounce-1..3 rays"),
                   "triangles": n_tris, "rays_per_gpu": n, "accel": args.accel.upper(), "traverse_mode": ("reference order, first hit ends the walk" if args.accel != "bvh" else "min t over the tree, best-t pruned")
                   if shadow else ("pruned" if args.accel != "grid" else "3D-DDA as written"),
                   "parallelism": "scene replica per GPU, rays sharded, no collective", "ray_binning": args.binning, "builder": args.builder, "hit_fraction": round(hit_frac, 4), "hits_checksum": checksum,
                   "scene_build_s": round(build_s, 2)},
        "roofline": {"bound": "hbm", "achieved": round(achieved_gbs, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(achieved_gbs / HBM_PEAK_GBS, 5), "traffic": None,
                     "kernel": {"bvh4": "k_bvh4_shadow<tri_only>" if shadow else "k_bvh4_closest<tri_only>",
                                "bvh": "k_bvh2<tri_only>", "grid": "k_grid<tri_only>"}[args.accel], "kernel_ms": round(avg_kernel_ms, 4), "binning_ms": round(float(np.mean(bin_ms)), 4),
                     "bytes_per_ray": round(bytes_per_ray, 2), "nodes_per_ray": round(nodes_per_ray, 3),
                     "tris_per_ray": round(tris_per_ray, 3)},
    }
    traffic_file = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(traffic_file):
        try:
            key = args.workload + ("@sah" if args.builder == "sah" else "") + ("/shadow" if shadow else "")
            tr = json.load(open(traffic_file)).get(key) if args.accel == "bvh4" else None
            if tr and tr.get("rays") == n and args.binning == "on":
                out["roofline"]["traffic"] = tr["hbm_bytes_per_launch"]
                # measured fabric bytes (PMC, profiles/traffic.json) over the live kernel time: the
                # caches serve part of the algorithmic bytes, which is how `frac` can pass 1
                out["roofline"]["traffic_gbs"] = round(tr["hbm_bytes_per_launch"] / (avg_kernel_ms * 1e-3) / 1e9, 1)
                out["roofline"]["traffic_frac"] = round(tr["hbm_bytes_per_launch"] / (avg_kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
                out["roofline"]["l2_hit_rate"] = tr.get("l2_hit_rate")
                if tr.get("tcp_lane_accesses_per_launch"):
                    # the measured limiter: the L1 serves 1.0 lane-divergent access per clock per CU
                    props = torch.cuda.get_device_properties(device)
                    clk_hz = float(getattr(props, "clock_rate", 2400000)) * 1e3
                    out["roofline"]["l1_lane_access_per_clk"] = round(
                        tr["tcp_lane_accesses_per_launch"] / (avg_kernel_ms * 1e-3) / clk_hz / props.multi_processor_count, 3)
                    out["roofline"]["l1_lane_access_peak"] = 1.0
        except Exception:
            pass

    if not shadow and args.accel == "bvh4":
        # secondary metric (BASELINE.json: "any-hit Mrays/s"), outside the timed region: shadow rays from
        # this batch's hit points toward a ceiling light, same scene, three launches
        s_rays, s_ld = gen_shadow_rays(sc, si, d_rays, 0xBEEF + rank, device)
        s_t = torch.empty(n, dtype=torch.float32, device=device)
        s_occ = torch.empty(n, dtype=torch.uint8, device=device)
        s_ms = []
        for _ in range(4):
            sc.shadow_dev(s_rays.data_ptr(), s_ld.data_ptr(), n, s_t.data_ptr(), s_occ.data_ptr(), stream)
            s_ms.append(sc.last_kernel_ms() + sc.last_binning_ms())
        s_best = shard.max_over_ranks(float(np.mean(s_ms[1:])), device)
        out["any_hit"] = {"metric": "Mrays/s any-hit (shadow rays toward an area light), QBVH",
                          "value": round(n * world / s_best / 1e3, 1), "unit": "Mrays/s", "ms_per_launch": round(s_best, 4),
                          "occluded_fraction": round(float(s_occ.float().mean().item()), 4)}
        del s_rays, s_ld, s_t, s_occ

    if not shadow and args.accel == "bvh4" and args.builder == "sah" and not args.no_reference_tree:
        # the same rays through the reference's own median-split QBVH (what `value` was measured on until the SAH builder
        # existed): rate, traversal statistics, and how many hit records differ between the two trees
        try:
            built_ref = api.BuiltScene(si, accel, sah=False)
            sc_ref = api.Scene(built_ref, local_rank)
            sc_ref.set_ray_binning(args.binning == "on")
            r_hits = torch.empty((n, 5), dtype=torch.int32, device=device)
            sc_ref.enable_counters(True)
            sc_ref.reset_counters()
            sc_ref.intersect_dev(d_rays.data_ptr(), n, r_hits.data_ptr(), stream)
            rc = sc_ref.counters()
            sc_ref.enable_counters(False)
            r_ms = []
            for _ in range(4):
                sc_ref.intersect_dev(d_rays.data_ptr(), n, r_hits.data_ptr(), stream)
                r_ms.append(sc_ref.last_kernel_ms() + sc_ref.last_binning_ms())
            r_best = shard.max_over_ranks(float(np.mean(r_ms[1:])), device)
            differ = (r_hits != d_hits).any(dim=1)
            ties = differ & (r_hits[:, 0] == d_hits[:, 0]) & (r_hits[:, 1] != d_hits[:, 1])      # bit-identical t, another object
            out["reference_tree"] = {"metric": "Mrays/s closest-hit (incoherent), QBVH built as the reference builds it (median split)",
                                     "value": round(n * world / r_best / 1e3, 1), "unit": "Mrays/s", "ms_per_launch": round(r_best, 4),
                                     "nodes_per_ray": round(rc["nodes_visited"] / n, 3), "tris_per_ray": round(rc["prims_tested"] / n, 3),
                                     "hit_records_differing_from_sah_tree": int(differ.sum().item()),
                                     "of_which_exact_t_ties": int(ties.sum().item())}
            del r_hits
            sc_ref.close()
            built_ref.close()
        except Exception as e:
            out["reference_tree"] = {"error": repr(e)[:300]}

    if not shadow and args.accel == "bvh4" and not args.no_render:
        # second half of BASELINE.json's metric: samples/sec/pixel of the wavefront path tracer on
        # the same courtyard opened to a constant sky, 4K, depth 5.  Sample passes are sharded over the ranks
        # (each renders its own sample range of the full frame); the colour buffers are summed over RCCL
        # inside the timed region.
        try:
            del d_rays, d_hits
            out["render"] = render_leg(args, api, scenes, shard, torch, rank, world, local_rank, device, barrier,
                                       n_copies, n_lat, n_lon)
        except Exception as e:  # the headline line must not depend on this leg
            out["render"] = {"error": repr(e)[:300]}

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        # the CPU checker timed as "the reference's path on the host cores": same tree, same rays
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import oraclelib as ol
        o = ol.OracleScene(si, accel, built=built)
        cores = min(os.cpu_count() or 1, int(os.environ.get("OMP_NUM_THREADS", "16")))
        host_rays = host_rays_keep
        host_ld = d_ld.cpu().numpy() if shadow else None

        def cpu(lo_, hi_, threads):
            if shadow:
                return o.shadow(host_rays[lo_:hi_], host_ld[lo_:hi_], nthreads=threads)
            return o.intersect(host_rays[lo_:hi_], mode=api.TRAVERSE_EXACT, nthreads=threads)

        t1 = time.perf_counter()
        cpu(0, 1 << 16, cores)
        rate = (1 << 16) / max(time.perf_counter() - t1, 1e-6)
        m = int(min(len(host_rays), max(1 << 16, rate * args.cpu_seconds)))
        t1 = time.perf_counter()
        ref = cpu(0, m, cores)
        cpu_s = time.perf_counter() - t1
        m1 = max(m // (2 * cores), 1)
        t1 = time.perf_counter()
        cpu(0, m1, 1)
        cpu1_s = time.perf_counter() - t1
        o0_mrays = None
        if os.path.exists(ol.ORACLE_O0_PATH):        # the reference's own Makefile builds without -O (Makefile:12)
            o_keep, o = o, ol.OracleScene(si, accel, built=built, lib_path=ol.ORACLE_O0_PATH)
            m0 = max(m1 // 4, 1)
            t1 = time.perf_counter()
            cpu(0, m0, 1)
            o0_mrays = round(m0 / (time.perf_counter() - t1) / 1e6, 4)
            o.close()
            o = o_keep
        if shadow:
            same = bool(ol.same_bits(d_t[:m].cpu().numpy(), ref[0])) and \
                bool(np.array_equal(d_occ[:m].cpu().numpy().astype(bool), ref[1]))
        else:
            same = bool(ol.same_bits(gpu_hits_host[:m], ref))
        out["cpu_baseline"] = {"value": round(m / cpu_s / 1e6, 4), "unit": "Mrays/s", "cores": cores, "kind": "port",
                               "sample": "first %d rays of the GPU batch, oracle in the reference's control flow, %.1f s" % (m, cpu_s),
                               "one_thread_mrays": round(m1 / cpu1_s / 1e6, 4), "one_thread_O0_mrays": o0_mrays,
                               "gpu_bit_identical_on_sample": same}
    if rank == 0:
        print(json.dumps(out), flush=True)
    sc.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
