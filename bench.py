#!/usr/bin/env python3
"""bench.py -- incoherent closest-hit throughput of the QBVH traversal kernel on MI355X.

  python bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path (n x intersectTest, /root/reference/intersect.h:365, BVH4
branch) over one batch of 2^24 incoherent rays that is already resident in HBM.  The workload is
BASELINE.json's headline configuration (configs[4]: "San Miguel ~10M tris, QBVH, incoherent diffuse
bounces") on the procedural stand-in scenes.courtyard (the real asset is not shipped and cannot be
fetched): 10 485 772 world-space triangles, QBVH built with the reference's rules, ray set
R-incoherent = shuffled bounce-1..3 cosine-hemisphere rays.

Multi-GPU: one process per GPU, full scene replica per GPU, no data-path collective (rays are
independent).  `--gpus N` with N > 1 and no WORLD_SIZE in the environment makes this process a launcher:
it starts N fresh rank processes (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT set) before
anything touches the GPU and relays rank 0's JSON line; under torchrun the ranks are already there.
Two ways of sharing the work, both in every N > 1 line:
  weak   (`value`, the contract's default) each rank traces its own 2^24-ray batch;
  strong one 2^24-ray batch cut into N contiguous slices (shard.shard_range), Mrays/s = 2^24 / max_i(time_i)
         (`--scaling strong` makes this one `value` and the weak figure the second key).
Timing = max over ranks, barrier + synchronize either side.

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

PG_TIMEOUT_S = 300             # process-group timeout: a collective that cannot complete fails well inside the driver's 600 s limit
HBM_PEAK_GBS = 8000.0          # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s (spec; ~6.3 TB/s achievable)
L2_PEAK_GBS = 34500.0          # same guide, "L2 (per XCD)": ~34.5 TB/s aggregate

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


def render_leg(args, api, scenes, shard, torch, rank, world, local_rank, device, red_device, barrier, n_copies, n_lat, n_lon):
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
    dt = shard.max_over_ranks(time.perf_counter() - t0, red_device)
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
           "reduce": ("RCCL all_reduce of the colour buffers" if args.dist_backend == "nccl" else "gloo all_reduce of host copies (rehearsal)")
           if world > 1 else "none (1 GPU)"}
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import oraclelib as ol
        cores = min(os.cpu_count() or 1, int(os.environ.get("OMP_NUM_THREADS", "16")))
        w2, h2, spp2 = 256, 144, args.render_cpu_spp       # sized for a few seconds of oracle time, not 0.2 s
        flm2 = api.make_film(cam, 1.0, w2, h2)
        o = ol.OracleScene(si, built=built)
        t1 = time.perf_counter()
        want = o.render(desc, cam, flm2, 0, spp2, nthreads=cores)
        cpu_s = time.perf_counter() - t1
        r2 = api.Renderer(sc, desc, cam, flm2)
        r2.render(spp2)
        got, _ = r2.read()
        r2.close()
        res["cpu_baseline"] = {"value": round(spp2 * w2 * h2 / cpu_s / (width * height), 5), "unit": "samples/s/pixel at %dx%d" % (width, height),
                               "cores": cores, "kind": "port", "sample": "%dx%d image, %d spp, oracle_render, %.1f s" % (w2, h2, spp2, cpu_s),
                               "gpu_image_bit_identical_on_sample": bool((got == want).all())}
        o.close()
    r.close()
    sc.close()
    built.close()
    return res


def src_hash():
    """Hash of the sources that decide the walk kernels' memory traffic: a profiles/traffic.json entry measured on
    other sources is stale and is not reported."""
    import hashlib
    h = hashlib.sha1()
    for f in ("crt_kernels.h", "crt_device.h", "crt_binning.h", "crt_build.cpp"):
        h.update(open(os.path.join(ROOT, "craytracer_amd", "csrc", f), "rb").read())
    return h.hexdigest()[:12]


_OUT_FD = 1


def emit(line):
    """the one JSON line, to the descriptor that was stdout when main() started"""
    sys.stdout.flush()
    os.write(_OUT_FD, (line + "\n").encode())


def spawn_ranks(n_gpus, argv, poll_s=0.2, grace_s=5.0):
    """--gpus N without a launcher: start N rank processes of this script (fresh interpreters: this process has
    not imported torch or touched HIP and never will), relay rank 0's output, and SUPERVISE them: the children are
    polled, and when one exits non-zero the others -- which would otherwise sit in a rendezvous or a collective
    until the process-group timeout -- are terminated (SIGTERM, then SIGKILL after `grace_s`) and the launcher
    exits non-zero with every return code.  BENCH_LAUNCH_TIMEOUT (seconds, default 540: below the driver's 600)
    bounds the whole run the same way.  Nothing is ever re-executed in a process that has touched the GPU."""
    import socket
    import subprocess
    import tempfile
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    cores = os.cpu_count() or 1
    if os.environ.get("OMP_NUM_THREADS"):
        cores = min(cores, int(os.environ["OMP_NUM_THREADS"]))
    per_rank = max(1, cores // n_gpus)
    out0 = tempfile.TemporaryFile()                     # a file, not a pipe: nobody has to drain it while we poll
    procs = []
    for r in range(n_gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n_gpus), LOCAL_WORLD_SIZE=str(n_gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("CRT_BUILD_THREADS", str(per_rank))        # host builder threads: this rank's share of the cores
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=out0 if r == 0 else subprocess.DEVNULL))
    deadline = time.time() + float(os.environ.get("BENCH_LAUNCH_TIMEOUT", "540"))
    why = None
    while True:
        rcs = [p.poll() for p in procs]
        if all(rc is not None for rc in rcs):
            break
        if any(rc not in (None, 0) for rc in rcs):
            why = "rank %d exited with code %d" % next((i, rc) for i, rc in enumerate(rcs) if rc not in (None, 0))
        elif time.time() > deadline:
            why = "BENCH_LAUNCH_TIMEOUT reached"
        if why:
            for p in procs:
                if p.poll() is None:
                    p.terminate()
            t_kill = time.time() + grace_s
            while time.time() < t_kill and any(p.poll() is None for p in procs):
                time.sleep(poll_s)
            for p in procs:
                if p.poll() is None:
                    p.kill()
            rcs = [p.wait() for p in procs]
            break
        time.sleep(poll_s)
    out0.seek(0)
    sys.stdout.write(out0.read().decode())
    sys.stdout.flush()
    if why or any(rcs):
        raise SystemExit("bench.py: %s; rank exit codes %s (the other ranks were stopped)" % (why or "a rank failed", rcs))


def dry_run(args, rank, world):
    """--dry-backend gloo: the multi-process plumbing of this script (rendezvous, slices, barriers, max-over-ranks
    timing, rank 0's single JSON line) with NO ray traced and no GPU touched -- the seam tests/test_bench_launch.py
    drives on CPU.  It measures nothing and says so."""
    import torch
    import torch.distributed as dist
    from craytracer_amd import shard
    if world > 1:
        import datetime
        dist.init_process_group(args.dry_backend, timeout=datetime.timedelta(seconds=PG_TIMEOUT_S))
    n = args.rays
    b, e = shard.shard_range(n, rank, world)
    slices = [None] * world
    if world > 1:
        dist.all_gather_object(slices, (rank, b, e))
        dist.barrier()
    else:
        slices = [(rank, b, e)]
    t0 = time.perf_counter()
    for _ in range(args.steps):
        time.sleep(0.001 * (rank + 1))
    if os.environ.get("BENCH_DRY_DIE_RANK") == str(rank):      # test seam: this rank dies mid-run, the others wait in the barrier
        os._exit(int(os.environ.get("BENCH_DRY_DIE_CODE", "3")))
    if world > 1:
        dist.barrier()
    elapsed = shard.max_over_ranks(time.perf_counter() - t0)
    buf = torch.full((4,), float(rank + 1))
    buf, spp = shard.combine_radiance(buf, 3)
    if rank == 0:
        emit(json.dumps({"metric": "dry run: launch plumbing only, nothing traced", "value": 0.0, "unit": "Mrays/s", "n_gpus": world,
                          "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / max(args.steps, 1) * 1e3, 4),
                          "higher_is_better": True, "scaling": args.scaling, "dry": True, "backend": args.dry_backend,
                          "slices": sorted(slices), "combined_radiance": buf.tolist(), "combined_spp": spp}))
    if world > 1:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="courtyard-10M", choices=sorted(WORKLOADS))
    ap.add_argument("--rays", type=int, default=1 << 24)
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="target wall time of the cpu_baseline sample")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--node-compression", default=None, choices=["on", "off"],
                    help="walk the 64-byte conservative node copy (default: the library's default)")
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
    ap.add_argument("--no-other-accels", action="store_true", help="skip the binary-BVH / uniform-grid figures of the default line")
    ap.add_argument("--no-render", action="store_true", help="skip the samples/sec/pixel leg (the wavefront path tracer)")
    ap.add_argument("--no-two-streams", action="store_true", help="skip the legs that enqueue the steps alternately on two streams (profiling "
                    "runs: overlapped launches of one kernel stretch each other, so per-kernel averages would not be the headline launch's)")
    ap.add_argument("--no-strong-projection", action="store_true", help="N = 1: skip the leg that traces the 8 shares of the batch one by one")
    ap.add_argument("--render-spp", type=int, default=32, help="samples per pixel per GPU timed in the render leg")
    ap.add_argument("--render-cpu-spp", type=int, default=48, help="samples per pixel of the render leg's 256x144 cpu_baseline image")
    ap.add_argument("--mode", default="closest", choices=["closest", "shadow"],
                    help="closest = the headline metric; shadow = the secondary any-hit metric")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="N > 1: weak = every rank its own batch (value), strong = one batch cut into N slices (value); the other one is "
                         "reported under its own key")
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"],
                    help="nccl (= RCCL, the default) or gloo: rehearsal of the N > 1 path on a box with fewer GPUs than ranks "
                         "(ranks share devices round-robin; RCCL refuses two ranks on one GPU, gloo reduces host copies)")
    ap.add_argument("--dry-backend", default=None, choices=["gloo"],
                    help="test seam: run the launch / rendezvous / slicing / timing plumbing over this CPU backend, trace nothing")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return spawn_ranks(args.gpus, sys.argv[1:])       # before anything imports torch / initialises HIP
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("bench.py: WORLD_SIZE=%d but --gpus %d (pass --gpus N under a launcher with N ranks)" % (world, args.gpus))
    # stdout carries rank 0's ONE JSON line and nothing else: whatever a library prints there on its own (gloo announces its
    # connections on stdout, from C) goes to stderr instead, and the line is written to the descriptor stdout had
    global _OUT_FD
    sys.stdout.flush()
    _OUT_FD = os.dup(1)
    os.dup2(2, 1)
    if args.dry_backend:
        return dry_run(args, rank, world)
    if world > 1 and "CRT_BUILD_THREADS" not in os.environ:
        # under a launcher of its own (torchrun) nobody has divided the host cores among the ranks: every rank builds its scene
        # replicas with its share of them (spawn_ranks sets the same variable for the ranks it starts)
        cores = os.cpu_count() or 1
        if os.environ.get("OMP_NUM_THREADS"):
            cores = min(cores, int(os.environ["OMP_NUM_THREADS"]))
        os.environ["CRT_BUILD_THREADS"] = str(max(1, cores // int(os.environ.get("LOCAL_WORLD_SIZE", world))))
    if args.builder is None:
        args.builder = "sah" if args.accel == "bvh4" else "reference"
    if args.builder == "sah" and args.accel != "bvh4":
        raise SystemExit("--builder sah is a QBVH option")

    if os.environ.get("BENCH_DUMP_AFTER"):              # debugging aid: every rank dumps its Python stack after that many seconds
        import faulthandler
        faulthandler.dump_traceback_later(float(os.environ["BENCH_DUMP_AFTER"]), exit=False)

    import torch
    from craytracer_amd import api, scenes, shard

    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: there is no CPU fallback for the traversal path")
    if args.dist_backend == "gloo":
        if world > 2 * torch.cuda.device_count():
            raise SystemExit("bench.py: --dist-backend gloo maps ranks onto devices round-robin and more than two rank processes "
                             "per GPU are known to stall in torch's own scan / sort kernels (DESIGN.md (d), \"A scare\"): "
                             "%d ranks on %d device(s) refused" % (world, torch.cuda.device_count()))
        local_rank = local_rank % torch.cuda.device_count()
    if world > 1:
        import datetime
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank), timeout=datetime.timedelta(seconds=PG_TIMEOUT_S))
        else:
            dist.init_process_group("gloo", timeout=datetime.timedelta(seconds=PG_TIMEOUT_S))
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    red_device = device if args.dist_backend == "nccl" else "cpu"     # where the timing scalars are reduced

    t_start = time.time()

    def progress(what):
        """one line on stderr per leg (rank 0): a harness that takes silence for a hang sees the run moving; stdout stays one JSON line"""
        if rank == 0:
            print("[bench %6.1f s] %s" % (time.time() - t_start, what), file=sys.stderr, flush=True)

    # ---- scene (full replica per GPU) ----
    t0 = time.time()
    n_copies, n_lat, n_lon = WORKLOADS[args.workload]
    si = scenes.courtyard(n_copies, n_lat, n_lon, seed=5)
    accel = {"bvh4": api.ACCEL_BVH4, "bvh": api.ACCEL_BVH, "grid": api.ACCEL_GRID}[args.accel]
    built = api.BuiltScene(si, accel, sah=(args.builder == "sah"))
    sc = api.Scene(built, local_rank)
    sc.set_ray_binning(args.binning == "on")
    if args.node_compression is not None:
        sc.set_node_compression(args.node_compression == "on")
    build_s = time.time() - t0
    n_tris = si.num_obj

    # ---- rays, resident in HBM.  `own` = this rank's batch (weak scaling; seed + rank); `shared` = the one batch every
    #      rank derives identically and traces a slice of (strong scaling).  At N = 1 they are the same batch. ----
    n = args.rays
    stream = torch.cuda.current_stream().cuda_stream
    shadow = args.mode == "shadow"
    progress("scene built and uploaded (%d triangles)" % n_tris)
    d_rays = gen_incoherent_rays(sc, si, n, 0xC0FFEE + rank, device)
    d_ld = None
    if shadow:
        d_rays, d_ld = gen_shadow_rays(sc, si, d_rays, 0xBEEF + rank, device)
    if world > 1 and rank != 0:
        d_shared = gen_incoherent_rays(sc, si, n, 0xC0FFEE, device)
        d_shared_ld = None
        if shadow:
            d_shared, d_shared_ld = gen_shadow_rays(sc, si, d_shared, 0xBEEF, device)
    else:
        d_shared, d_shared_ld = d_rays, d_ld
    lo_s, hi_s = shard.shard_range(n, rank, world)                 # this rank's slice of the shared batch
    d_hits = torch.empty((n, 5), dtype=torch.int32, device=device)
    d_t = torch.empty(n, dtype=torch.float32, device=device) if shadow else None
    d_occ = torch.empty(n, dtype=torch.uint8, device=device) if shadow else None

    def trace(rays, ld, first, count):
        """one pass of the hot path over rays[first : first + count] (device pointers into the resident batch)"""
        if count == 0:
            return
        rp = rays.data_ptr() + first * 24
        if shadow:
            sc.shadow_dev(rp, ld.data_ptr() + first * 4, count, d_t.data_ptr() + first * 4, d_occ.data_ptr() + first, stream)
        else:
            sc.intersect_dev(rp, count, d_hits.data_ptr() + first * 20, stream)

    def step_weak():
        trace(d_rays, d_ld, 0, n)

    def trace_slice(rays, ld, k, count, st=None, which=0):
        """slice k of `count` of the whole resident batch: the library bins the batch and walks a run of whole cells
        (crt_*_batch_dev_slice); results land at the rays' own indices"""
        st = stream if st is None else st
        if shadow:
            sc.shadow_dev_slice(rays.data_ptr(), ld.data_ptr(), n, d_t.data_ptr(), d_occ.data_ptr(), k, count, st)
        else:
            sc.intersect_dev_slice(rays.data_ptr(), n, (d_hits_b if which else d_hits).data_ptr(), k, count, st)

    def step_strong():
        trace_slice(d_shared, d_shared_ld, rank, world)

    # Two streams: consecutive steps are independent launches; enqueued alternately on two streams they overlap where the
    # library allows it (a scene holds two sets of launch scratch) -- the drain tail of one step, its longest rays walking on
    # alone, runs beside the start of the next.  Every step is complete when the timed region ends.  Closest-hit steps write
    # alternate result buffers; any-hit steps the same one (identical values).
    side = [torch.cuda.Stream(device=device) for _ in range(2)]
    d_hits_b = torch.empty((n, 5), dtype=torch.int32, device=device) if not shadow else None

    def timed2(step_fn, steps, warmup):
        """as timed(), steps enqueued alternately on two streams: step_fn(stream pointer, 0 | 1)"""
        for i in range(warmup):
            step_fn(side[i & 1].cuda_stream, i & 1)
        barrier()
        t_start = time.perf_counter()
        for i in range(steps):
            step_fn(side[i & 1].cuda_stream, i & 1)
        barrier()
        return shard.max_over_ranks(time.perf_counter() - t_start, red_device)

    def step_weak2(st, which):
        if shadow:
            sc.shadow_dev(d_rays.data_ptr(), d_ld.data_ptr(), n, d_t.data_ptr(), d_occ.data_ptr(), st)
        else:
            sc.intersect_dev(d_rays.data_ptr(), n, (d_hits_b if which else d_hits).data_ptr(), st)

    def step_strong2(st, which):
        trace_slice(d_shared, d_shared_ld, rank, world, st, which)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(step_fn, steps, warmup, with_kernel_ms):
        """`warmup` untimed steps, then exactly `steps` steps between barrier + synchronize; max over ranks."""
        for _ in range(warmup):
            step_fn()
        barrier()
        k_ms, b_ms = [], []
        t_start = time.perf_counter()
        for _ in range(steps):
            step_fn()
            if with_kernel_ms:
                k_ms.append(sc.last_kernel_ms())       # HIP events on the launch stream (syncs the step)
                b_ms.append(sc.last_binning_ms())
        barrier()
        return shard.max_over_ranks(time.perf_counter() - t_start, red_device), k_ms, b_ms

    # per-ray traversal statistics (defines the algorithmic bytes), outside the timed region
    sc.enable_counters(True)
    sc.reset_counters()
    step_weak()
    cnt = sc.counters()
    sc.enable_counters(False)
    nodes_per_ray = cnt["nodes_visited"] / n
    tris_per_ray = cnt["prims_tested"] / n
    io_bytes = 28 + 5 if shadow else 24 + 20          # shadow: ray + light_dist in, t + occluded out

    compressed_nodes = args.accel == "bvh4" and args.node_compression != "off"

    def algorithmic_bytes(nodes, tris):
        """SURVEY.md 8(d), NO-CACHE model: every node visit and every triangle test charged as if it came from HBM.
        24 (ray in) + 20 (hit out) + one QBVH node per visit + 36 + 4 per triangle tested; a node visit reads 112 of
        the node's 128 bytes (6 plane rows + the child words), so 112 is charged, not SURVEY's planning value 128."""
        if args.accel == "bvh4":                       # 64: one record of the conservative node copy (the default walk)
            return io_bytes + (64 if compressed_nodes else 112) * nodes + 40 * tris
        if args.accel == "bvh":                        # 32-byte nodes
            return io_bytes + 32 * nodes + 40 * tris
        return io_bytes + 8 * nodes + 44 * tris        # grid: nodes = cells walked, 8 B of cell_start; + 4 B item index per object

    bytes_per_ray = algorithmic_bytes(nodes_per_ray, tris_per_ray)

    strong_is_value = args.scaling == "strong" and world > 1
    progress("rays generated, counters taken; timing the headline leg")
    elapsed, kernel_ms, bin_ms = timed(step_strong if strong_is_value else step_weak, args.steps, args.warmup, True)
    rays_per_step = float(n) if strong_is_value else float(n) * world
    value = rays_per_step * args.steps / elapsed / 1e6
    other = None
    if world > 1:                                      # the other way of sharing the work: fewer launches, own key
        o_steps = max(2, min(args.steps, 6))
        o_el, _, _ = timed(step_weak if strong_is_value else step_strong, o_steps, 1, False)
        o_rays = float(n) * world if strong_is_value else float(n)
        other = {"scaling": "weak" if strong_is_value else "strong", "value": round(o_rays * o_steps / o_el / 1e6, 3), "unit": "Mrays/s",
                 "steps": o_steps, "ms_per_step": round(o_el / o_steps * 1e3, 4),
                 "rays_per_step_all_gpus": int(o_rays),
                 "sharding": "one batch, rank k walks share k of %d: whole binning cells in Morton order (crt_intersect_batch_dev_slice)" % world
                 if not strong_is_value else "every rank its own batch"}
    # the same steps enqueued alternately on two streams (both ways of sharing the work)
    two = {}
    p_steps = max(4, min(args.steps, 10))
    if not args.no_two_streams:
        p_el = timed2(step_weak2, p_steps, 2)
        two["weak"] = {"value": round(float(n) * world * p_steps / p_el / 1e6, 3), "unit": "Mrays/s", "steps": p_steps, "ms_per_step": round(p_el / p_steps * 1e3, 4)}
    if world > 1 and not args.no_two_streams:
        p_el = timed2(step_strong2, p_steps, 2)
        two["strong"] = {"value": round(float(n) * p_steps / p_el / 1e6, 3), "unit": "Mrays/s", "steps": p_steps, "ms_per_step": round(p_el / p_steps * 1e3, 4)}
    step_weak()                                        # leave d_hits holding this rank's own batch for the checks below
    torch.cuda.synchronize()
    if shadow:
        hit_frac = float(d_occ.float().mean().item())
        checksum = int(d_t.view(torch.int32).to(torch.int64).sum().item() + d_occ.to(torch.int64).sum().item()) & 0xFFFFFFFFFFFF
    else:
        hit_frac = float((d_hits[:, 1] >= 0).float().mean().item())
        checksum = int(d_hits.to(torch.int64).sum().item()) & 0xFFFFFFFFFFFF  # same scene + rays => same number

    gpu_hits_host = None if shadow else d_hits.cpu().numpy().view(api.HIT_DTYPE).reshape(-1)
    host_rays_keep = d_rays.cpu().numpy()
    avg_kernel_ms = float(np.mean(kernel_ms))
    rays_per_launch = (hi_s - lo_s) if strong_is_value else n
    props = torch.cuda.get_device_properties(device)
    clk_hz = float(getattr(props, "clock_rate", 2400000)) * 1e3
    shash = src_hash()
    traffic_db = {}
    try:
        traffic_db = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
    except Exception:
        pass

    def roofline(kernel_name, kernel_ms_, binning_ms_, nodes, tris, key, chk, n_launch):
        """HBM roofline of one walk-kernel launch.  `achieved` / `frac` are the COUNTER-based HBM-side figure (PMC passes of
        this same command, profiles/traffic.json; an entry counts only if it was measured on these sources, this workload and
        these hit records -- else traffic is null and so is frac).  The SURVEY 8(d) algorithmic bytes are reported beside it
        under `algorithmic`, labelled as the no-cache model they are: L2 serves about half the lines, so that figure divided by
        the kernel time is NOT a rate any memory level sustains and is not compared with a peak."""
        bpr = algorithmic_bytes(nodes, tris)
        rf = {"bound": "hbm", "achieved": None, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": None, "traffic": None,
              "kernel": kernel_name, "kernel_ms": round(kernel_ms_, 4), "binning_ms": round(binning_ms_, 4),
              "algorithmic": {"model": "no-cache", "bytes_per_ray": round(bpr, 2), "bytes_per_launch": int(bpr * n_launch),
                              "gbs_if_nothing_were_cached": round(bpr * n_launch / (kernel_ms_ * 1e-3) / 1e9, 1),
                              "nodes_per_ray": round(nodes, 3), "tris_per_ray": round(tris, 3)}}
        tr = traffic_db.get(key)
        match = None
        if tr and tr.get("rays") == n_launch and tr.get("src_hash") == shash and args.binning == "on":
            if tr.get("hits_checksum") == chk:
                match = "hit-record checksum"
            elif tr.get("nodes_per_ray") and abs(tr["nodes_per_ray"] / nodes - 1) < 1e-4 and abs(tr["tris_per_ray"] / tris - 1) < 1e-4:
                match = "nodes and triangles per ray to 1e-4 (ray batch differs in the few rays that leave exact-t ties)"
        if match:
            rf["traffic_matched_by"] = match
            t = tr["hbm_bytes_per_launch"]
            gbs = t / (kernel_ms_ * 1e-3) / 1e9
            rf.update({"achieved": round(gbs, 1), "frac": round(gbs / HBM_PEAK_GBS, 4), "traffic": t,
                       "traffic_over_algorithmic": round(t / (bpr * n_launch), 3), "traffic_source": tr.get("source"),
                       "hbm_achievable_gbs": 6300.0, "frac_of_achievable": round(gbs / 6300.0, 4)})
            if tr.get("l2_requests_per_launch"):
                l2 = tr["l2_requests_per_launch"] * 128.0 / (kernel_ms_ * 1e-3) / 1e9
                rf["l2"] = {"achieved": round(l2, 1), "peak": L2_PEAK_GBS, "unit": "GB/s", "frac": round(l2 / L2_PEAK_GBS, 4),
                            "hit_rate": tr.get("l2_hit_rate"), "requests_per_launch": tr["l2_requests_per_launch"]}
            if tr.get("tcp_lane_accesses_per_launch"):
                # per-CU L1 (TCP) access rate against the 1.0 per clock per CU that tools/calib_tcp.hip measures for this pattern
                rf["l1_lane_access_per_clk"] = round(tr["tcp_lane_accesses_per_launch"] / (kernel_ms_ * 1e-3) / clk_hz / props.multi_processor_count, 3)
                rf["l1_lane_access_peak"] = 1.0
            for k_ in ("active_lane_fraction", "valu_issue_utilisation", "waves_per_simd"):
                if tr.get(k_) is not None:
                    rf[k_] = tr[k_]
            rf["frac_is"] = ("fabric-side bytes (what leaves the L2s: 2 x FETCH_SIZE + WRITE_SIZE, Infinity-Cache hits included -- rocprofv3 on "
                             "gfx950 offers no counter behind the Infinity Cache) / kernel time / HBM peak: an upper bound of the HBM fraction")
            if tr.get("ea_read_latency_clk") is not None:
                # What binds, from the counters of the same passes: the reads that leave the L2s take `ea_read_latency_clk` on average and
                # the L2 channels are (almost) never out of DRAM credits, i.e. the memory side is not pushing back; the traffic is the
                # reads in flight x 128 B / that latency (Little's law).  The walk is a chain of dependent fetch -> test -> decide steps
                # whose concurrency (5 waves per SIMD, one fetch group per lane and iteration) fixes that number: "latency", priced
                # against the HBM peak because the contract's roofline classes are hbm | mfma and this path has no contraction.
                lat, infl, stall = tr["ea_read_latency_clk"], tr.get("ea_reads_in_flight"), tr.get("dram_credit_stall_fraction")
                rf["bound_evidence"] = {"ea_read_latency_l2_clk": lat, "ea_reads_in_flight": infl, "dram_credit_stall_fraction": stall,
                                        "wave_cycles_waiting": tr.get("wave_cycles_waiting"),
                                        "valu_issue_utilisation": tr.get("valu_issue_utilisation")}
                if stall is not None and stall < 0.02:
                    rf["bound"] = "latency"
                    rf["roofline_class"] = "hbm"
        return rf

    accel_name = {"bvh4": "QBVH", "bvh": "binary BVH", "grid": "uniform grid"}[args.accel]
    tree = ("SAH split planes (crt_build_scene_ex CRT_BUILD_SAH: the reference's node format, leaf rule and walk; not the tree the "
            "reference's builder makes)" if args.builder == "sah" else "the reference's median split (bvh4.h:127-297)")
    kname = {"bvh4": "k_bvh4_shadow<tri_only>" if shadow else "k_bvh4_closest_tri", "bvh": "k_bvh2<tri_only>", "grid": "k_grid<tri_only>"}[args.accel]
    tkey = "%s@%s%s" % (args.workload, args.builder, "/shadow" if shadow else "")
    out = {
        "metric": ("Mrays/s any-hit (shadow rays toward an area light), %s" if shadow else "Mrays/s closest-hit (incoherent), %s") % accel_name,
        "value": round(value, 3), "unit": "Mrays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "strong" if strong_is_value else "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "%s: %d-triangle procedural courtyard (San Miguel stand-in), %s, %d %s per %s per step"
                               % (args.workload, n_tris, accel_name + (" built with " + tree if args.accel == "bvh4" else ""), n,
                                  "shadow rays from the incoherent hit points toward a ceiling light" if shadow else "incoherent bounce-1..3 rays",
                                  "job (cut into %d slices)" % world if strong_is_value else "GPU"),
                   "triangles": n_tris, "rays_per_gpu": rays_per_launch, "accel": args.accel.upper(),
                   "tree": "sah" if args.builder == "sah" else "reference",
                   "traverse_mode": ("reference order, first hit ends the walk" if args.accel != "bvh" else "min t over the tree, best-t pruned")
                   if shadow else ("pruned" if args.accel != "grid" else "3D-DDA as written"),
                   "parallelism": "scene replica per GPU, rays sharded, no collective", "ray_binning": args.binning, "builder": args.builder,
                   "node_format": {"bvh": "64-byte records of both child boxes (4 x 16 B per visit), nearer child first",
                                   "grid": "cell_start[] pairs + cell_items[]"}.get(args.accel)
                   or ("64-byte conservative copy (4 x 16 B per visit)" if compressed_nodes else "128-byte nodes (7 x 16 B per visit)"),
                   "hit_fraction": round(hit_frac, 4), "hits_checksum": checksum, "src_hash": shash, "scene_build_s": round(build_s, 2)},
        "roofline": roofline(kname, avg_kernel_ms, float(np.mean(bin_ms)), nodes_per_ray, tris_per_ray, tkey, checksum, rays_per_launch),
    }
    if other:
        out[other["scaling"] + "_scaling"] = other
    if two:
        out["two_streams"] = dict(two, what="the same steps enqueued alternately on two HIP streams: consecutive launches overlap (the drain "
                                           "tail of one beside the start of the next); not the headline, whose steps run on one stream")

    progress("headline leg done")
    if world == 1 and args.accel == "bvh4" and not shadow and not args.no_strong_projection and two:
        # What one GPU can say about strong scaling (ONE batch cut into K shares, Mrays/s(K) = n / max_k time_k): each of the
        # K = 8 shares of THIS batch traced on its own, as rank k of 8 would (crt_intersect_batch_dev_slice: the whole batch binned, a run
        # of whole cells walked), wall-clock with the same barriers as the headline.  The shares must tile the batch: every record of
        # the eight launches together is compared with the full launch's.
        K = 8
        full = d_hits.clone()
        d_hits.fill_(0x7F7F7F7F)
        wall_ms, wall2_ms, kern_ms, bin_ms_k, rays_k, done = [], [], [], [], [], 0
        for k in range(K):
            el, kms, bms = timed(lambda: trace_slice(d_rays, d_ld, k, K), 5, 1, True)
            wall_ms.append(el / 5 * 1e3)
            wall2_ms.append(timed2(lambda st, which: trace_slice(d_rays, d_ld, k, K, st, 0), 8, 2) / 8 * 1e3)
            kern_ms.append(float(np.mean(kms)))
            bin_ms_k.append(float(np.mean(bms)))
            now = int((d_hits[:, 1] != 0x7F7F7F7F).sum().item())
            rays_k.append(now - done)
            done = now
        tiles = bool((d_hits == full).all().item())
        single = elapsed / args.steps * 1e3
        out["strong_scaling_projection"] = {
            "what": "one GPU, the K shares of the headline batch traced one after the other; speedup = full-batch ms / slowest share's ms",
            "slices": K, "sharding": "whole binning cells in Morton order (crt_intersect_batch_dev_slice)",
            "full_batch_ms": round(single, 4), "share_wall_ms": [round(x, 4) for x in wall_ms], "share_walk_kernel_ms": [round(x, 4) for x in kern_ms],
            "share_binning_ms": [round(x, 4) for x in bin_ms_k], "share_rays": rays_k,
            "projected_speedup_at_%d_gpus" % K: round(single / max(wall_ms), 3),
            "two_streams": {"full_batch_ms": two["weak"]["ms_per_step"], "share_wall_ms": [round(x, 4) for x in wall2_ms],
                            "projected_speedup_at_%d_gpus" % K: round(two["weak"]["ms_per_step"] / max(wall2_ms), 3)},
            "shares_tile_the_batch_bit_identically": tiles}
        del full
        progress("strong-scaling projection done")
    if not shadow and args.accel == "bvh4":
        # secondary metric (BASELINE.json: "any-hit Mrays/s"), outside the timed region: shadow rays from
        # this batch's hit points toward a ceiling light, same scene, three launches
        s_rays, s_ld = gen_shadow_rays(sc, si, d_rays, 0xBEEF + rank, device)
        s_t = torch.empty(n, dtype=torch.float32, device=device)
        s_occ = torch.empty(n, dtype=torch.uint8, device=device)
        s_ms = []
        barrier()
        for _ in range(4):
            sc.shadow_dev(s_rays.data_ptr(), s_ld.data_ptr(), n, s_t.data_ptr(), s_occ.data_ptr(), stream)
            s_ms.append(sc.last_kernel_ms() + sc.last_binning_ms())
        s_best = shard.max_over_ranks(float(np.mean(s_ms[1:])), red_device)
        out["any_hit"] = {"metric": "Mrays/s any-hit (shadow rays toward an area light), QBVH",
                          "value": round(n * world / s_best / 1e3, 1), "unit": "Mrays/s", "ms_per_launch": round(s_best, 4),
                          "occluded_fraction": round(float(s_occ.float().mean().item()), 4)}
        del s_rays, s_ld, s_t, s_occ

    progress("any-hit leg done")
    if not shadow and args.accel == "bvh4" and args.builder == "sah" and not args.no_reference_tree:
        # the same rays through the tree the reference's own builder makes (integration route (b): the host keeps BVH4_build):
        # rate, its own roofline block, and how many hit records differ between the two trees
        try:
            built_ref = api.BuiltScene(si, accel, sah=False)
            sc_ref = api.Scene(built_ref, local_rank)
            sc_ref.set_ray_binning(args.binning == "on")
            if args.node_compression is not None:
                sc_ref.set_node_compression(args.node_compression == "on")
            r_hits = torch.empty((n, 5), dtype=torch.int32, device=device)
            sc_ref.enable_counters(True)
            sc_ref.reset_counters()
            sc_ref.intersect_dev(d_rays.data_ptr(), n, r_hits.data_ptr(), stream)
            rc = sc_ref.counters()
            sc_ref.enable_counters(False)
            r_ms, r_kms, r_bms = [], [], []
            barrier()
            for _ in range(4):
                sc_ref.intersect_dev(d_rays.data_ptr(), n, r_hits.data_ptr(), stream)
                r_kms.append(sc_ref.last_kernel_ms())
                r_bms.append(sc_ref.last_binning_ms())
                r_ms.append(r_kms[-1] + r_bms[-1])
            r_best = shard.max_over_ranks(float(np.mean(r_ms[1:])), red_device)
            differ = (r_hits != d_hits).any(dim=1)
            ties = differ & (r_hits[:, 0] == d_hits[:, 0]) & (r_hits[:, 1] != d_hits[:, 1])      # bit-identical t, another object
            r_chk = int(r_hits.to(torch.int64).sum().item()) & 0xFFFFFFFFFFFF
            out["reference_tree"] = {"metric": "Mrays/s closest-hit (incoherent), QBVH built as the reference builds it (median split)",
                                     "value": round(n * world / r_best / 1e3, 1), "unit": "Mrays/s", "ms_per_launch": round(r_best, 4),
                                     "hits_checksum": r_chk,
                                     "hit_records_differing_from_sah_tree": int(differ.sum().item()),
                                     "of_which_exact_t_ties": int(ties.sum().item()),
                                     "roofline": roofline("k_bvh4_closest_tri", float(np.mean(r_kms[1:])), float(np.mean(r_bms[1:])),
                                                          rc["nodes_visited"] / n, rc["prims_tested"] / n,
                                                          "%s@reference" % args.workload, r_chk, n)}
            del r_hits
            sc_ref.close()
            built_ref.close()
        except Exception as e:
            out["reference_tree"] = {"error": repr(e)[:300]}

    progress("reference-tree leg done")
    if not shadow and args.accel == "bvh4" and not args.no_render:
        # second half of BASELINE.json's metric: samples/sec/pixel of the wavefront path tracer on
        # the same courtyard opened to a constant sky, 4K, depth 5.  Sample passes are sharded over the ranks
        # (each renders its own sample range of the full frame); the colour buffers are summed over RCCL
        # inside the timed region.
        try:
            del d_rays, d_hits, d_shared, d_hits_b
            out["render"] = render_leg(args, api, scenes, shard, torch, rank, world, local_rank, device, red_device, barrier,
                                       n_copies, n_lat, n_lon)
        except Exception as e:  # the headline line must not depend on this leg
            out["render"] = {"error": repr(e)[:300]}

    progress("render leg done")
    if not shadow and args.accel == "bvh4" and world == 1 and not args.no_other_accels:
        # the reference's other two accelerators on the same kind of rays: extra figures, not the headline.
        # Each scene makes its own incoherent batch (bounce rays traced through that scene), as `--accel bvh|grid` does.
        others = {}
        for name, a_id, wl in (("binary_bvh", api.ACCEL_BVH, args.workload), ("uniform_grid", api.ACCEL_GRID, "courtyard-1M")):
            try:
                o_si = si if wl == args.workload else scenes.courtyard(*WORKLOADS[wl], seed=5)
                o_built = api.BuiltScene(o_si, a_id)
                o_sc = api.Scene(o_built, local_rank)
                o_rays = gen_incoherent_rays(o_sc, o_si, n, 0xC0FFEE + rank, device)
                o_hits = torch.empty((n, 5), dtype=torch.int32, device=device)
                ms = []
                for _ in range(4):
                    o_sc.intersect_dev(o_rays.data_ptr(), n, o_hits.data_ptr(), stream)
                    ms.append(o_sc.last_kernel_ms() + o_sc.last_binning_ms())
                c_ms = float(np.mean(ms[1:]))
                s_rays, s_ld = gen_shadow_rays(o_sc, o_si, o_rays, 0xBEEF + rank, device)
                s_t = torch.empty(n, dtype=torch.float32, device=device)
                s_occ = torch.empty(n, dtype=torch.uint8, device=device)
                ms = []
                for _ in range(4):
                    o_sc.shadow_dev(s_rays.data_ptr(), s_ld.data_ptr(), n, s_t.data_ptr(), s_occ.data_ptr(), stream)
                    ms.append(o_sc.last_kernel_ms() + o_sc.last_binning_ms())
                a_ms = float(np.mean(ms[1:]))
                others[name] = {"workload": wl, "triangles": o_si.num_obj, "rays": n,
                                "closest_hit_mrays": round(n / c_ms / 1e3, 1), "any_hit_mrays": round(n / a_ms / 1e3, 1),
                                "hits_checksum": int(o_hits.to(torch.int64).sum().item()) & 0xFFFFFFFFFFFF}
                del o_rays, o_hits, s_rays, s_ld, s_t, s_occ
                o_sc.close()
                o_built.close()
            except Exception as e:
                others[name] = {"error": repr(e)[:300]}
        out["other_accelerators"] = others

    progress("other accelerators done")
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
        emit(json.dumps(out))
    sc.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
