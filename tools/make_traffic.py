#!/usr/bin/env python3
"""gpurun_out/<dir> written by tools/prof_traffic.sh  ->  profiles/traffic.json entries + profiles/<tag>_pmc_<builder>_<mode>.txt.

Each entry is keyed `<workload>@<builder>[/shadow]` and carries what bench.py checks before it reports the number: the
ray count, the hit-record checksum of the profiled run and the hash of the kernel / builder sources.  Bytes:
read = 2 x FETCH_SIZE KB (gfx950 tallies 128-byte requests at 64 B: MI355X_MICROARCH.md "HBM", confirmed for this access
pattern by tools/calib_fetch.hip) cross-checked against TCC_EA0_RDREQ x 128 B; write = WRITE_SIZE KB."""
import json
import os
import re
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, tag = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "r03")
path = os.path.join(ROOT, "profiles", "traffic.json")
db = json.load(open(path)) if os.path.exists(path) else {}
for d in sorted(os.listdir(src)):
    full = os.path.join(src, d)
    if not os.path.isdir(full) or "_" not in d or not os.path.exists(os.path.join(full, "summary.txt")) \
            or not os.path.exists(os.path.join(full, "bench_line.json")):
        continue
    builder, mode = d.rsplit("_", 1)
    line = json.loads(open(os.path.join(full, "bench_line.json")).read())
    want = "k_bvh4_shadow<true, false" if mode == "shadow" else "k_bvh4_closest_tri<false, false>"    # the plain walk, not the tail-splitting instance of small launches
    vals, cur = {}, None
    for ln in open(os.path.join(full, "summary.txt")):
        if ln.startswith("kernel:"):
            cur = ln.split("kernel:", 1)[1].strip()
        else:
            m = re.match(r"\s+(\S+)\s+dispatches=(\d+) mean=(\S+) min=(\S+)", ln)
            if m and cur and want in cur:
                # SQ_WAVES: the smallest dispatch (a counter pass occasionally tallies one dispatch twice over); everything else: the mean
                vals[m.group(1)] = float(m.group(4) if m.group(1) == "SQ_WAVES" else m.group(3))
    if "FETCH_SIZE" not in vals:
        print("skip", d, "(kernel %s not found)" % want)
        continue
    read_b = 2.0 * vals["FETCH_SIZE"] * 1024.0
    write_b = vals.get("WRITE_SIZE", 0.0) * 1024.0
    n = line["config"]["rays_per_gpu"]
    workload = line["config"]["workload"].split(":")[0]
    key = "%s@%s%s" % (workload, builder, "/shadow" if mode == "shadow" else "")
    alg = line["roofline"]["algorithmic"]
    e = {"rays": n, "hits_checksum": line["config"]["hits_checksum"], "src_hash": line["config"]["src_hash"],
         "nodes_per_ray": alg["nodes_per_ray"], "tris_per_ray": alg["tris_per_ray"],
         "hbm_bytes_per_launch": int(read_b + write_b), "read_bytes": int(read_b), "write_bytes": int(write_b),
         "read_bytes_from_rdreq": int(vals.get("TCC_EA0_RDREQ_sum", 0) * 128),
         "l2_requests_per_launch": int(vals.get("TCC_REQ_sum", 0)),
         "l2_hit_rate": round(vals["TCC_HIT_sum"] / (vals["TCC_HIT_sum"] + vals["TCC_MISS_sum"]), 4) if "TCC_HIT_sum" in vals else None,
         "tcp_lane_accesses_per_launch": vals.get("TCP_TOTAL_CACHE_ACCESSES_sum"),
         "source": "profiles/%s_pmc_%s_%s.txt" % (tag, builder, mode)}
    if "SQ_THREAD_CYCLES_VALU" in vals and vals.get("SQ_ACTIVE_INST_VALU"):
        e["active_lane_fraction"] = round(vals["SQ_THREAD_CYCLES_VALU"] / (64.0 * vals["SQ_ACTIVE_INST_VALU"]), 4)
    if "SQ_BUSY_CYCLES" in vals and vals.get("SQ_ACTIVE_INST_VALU"):
        e["valu_instructions"] = vals.get("SQ_INSTS_VALU")
        e["salu_instructions"] = vals.get("SQ_INSTS_SALU")
    if vals.get("GRBM_GUI_ACTIVE") and vals.get("SQ_ACTIVE_INST_VALU"):
        # a wave64 VALU instruction occupies its SIMD's issue port for 4 cycles; GRBM_GUI_ACTIVE is summed over the 8 XCDs; 1 024 SIMDs
        e["valu_issue_utilisation"] = round(4.0 * vals["SQ_ACTIVE_INST_VALU"] / (1024.0 * vals["GRBM_GUI_ACTIVE"] / 8.0), 4)
    if "SQ_WAVES" in vals:
        e["waves_per_simd"] = round(vals["SQ_WAVES"] / 1024.0, 2)
    if vals.get("TCC_EA0_RDREQ_LEVEL_sum") and vals.get("TCC_EA0_RDREQ_sum"):
        # rocprofv3 --list-avail: "average read latency = EA_RDREQ_LEVEL / EA_RDREQ", in L2 clocks, of the reads that leave the L2s
        e["ea_read_latency_clk"] = round(vals["TCC_EA0_RDREQ_LEVEL_sum"] / vals["TCC_EA0_RDREQ_sum"], 1)
        if vals.get("GRBM_GUI_ACTIVE"):
            # reads in flight past the L2s, averaged over the launch (Little's law: LEVEL summed over cycles / cycles of one XCD)
            e["ea_reads_in_flight"] = round(vals["TCC_EA0_RDREQ_LEVEL_sum"] / (vals["GRBM_GUI_ACTIVE"] / 8.0), 1)
            if vals.get("TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum") is not None:
                # cycles (summed over the 128 L2 channels) in which a channel's read interface was out of DRAM credits
                e["dram_credit_stall_fraction"] = round(vals["TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum"] / (128.0 * vals["GRBM_GUI_ACTIVE"] / 8.0), 4)
    if vals.get("SQ_WAIT_ANY") and vals.get("SQ_WAVE_CYCLES"):
        e["wave_cycles_waiting"] = round(vals["SQ_WAIT_ANY"] / vals["SQ_WAVE_CYCLES"], 3)
    db[key] = e
    shutil.copy(os.path.join(full, "summary.txt"), os.path.join(ROOT, e["source"]))
    print(key, json.dumps(e))
db = {k: v for k, v in db.items() if k.startswith("_") or (isinstance(v, dict) and "src_hash" in v)}     # entries of older formats go
db["_how"] = ("rocprofv3 --pmc passes of `python3 bench.py --no-cpu-baseline --no-reference-tree --no-render --steps 5 --warmup 1 --builder B "
              "--mode M` (tools/prof_traffic.sh, one counter group per run), mean per launch of the timed walk-kernel instantiation; "
              "tools/make_traffic.py turns the summaries into these entries. read = 2 x FETCH_SIZE KB (gfx950 tallies 128-B requests at 64 B; "
              "confirmed by tools/calib_fetch.hip, profiles/r01_fetch_size_calibration.txt), write = WRITE_SIZE KB. Infinity-Cache hits are "
              "included in these counters, so true HBM traffic is at most this. An entry is used by bench.py only when rays and src_hash equal the live run's and "
              "either the hit-record checksum does too or (the reference-tree leg of the default line, whose ray batch was generated through "
              "the SAH tree and so differs from `--builder reference`'s in the handful of rays that leave exact-t ties) nodes / triangles per "
              "ray agree to 1e-4.")
json.dump(db, open(path, "w"), indent=1)
