#!/usr/bin/env python3
"""gpurun_out/<tag>_* written by tools/final_evidence.sh -> profiles/ (traffic.json + PMC summaries through make_traffic.py, the
binary-BVH counters, the default bench line, the kernel-trace stats) under the names profiles/README.md lists.
    python tools/final_evidence_collect.py <tag>"""
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
out = os.path.join(ROOT, "gpurun_out")
prof = os.path.join(ROOT, "profiles")
subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "make_traffic.py"), os.path.join(out, tag + "_traffic")], stdout=subprocess.DEVNULL)
shutil.copy(os.path.join(out, tag + "_pmc_bvh", "summary.txt"), os.path.join(prof, tag + "_pmc_binary_bvh_conservative.txt"))
shutil.copy(os.path.join(out, tag + "_bench.json"), os.path.join(prof, tag + "_bench_line_default.json"))
for n in ("full", "walk", "walk_with_reference_tree", "bvh", "grid"):
    shutil.copy(os.path.join(out, tag + "_stats", "kernel_stats_%s.csv" % n), os.path.join(prof, "%s_kernel_stats_%s.csv" % (tag, n)))
    shutil.copy(os.path.join(out, tag + "_stats", "bench_line_%s.json" % n), os.path.join(prof, "%s_bench_line_%s.json" % (tag, n)))
print("collected", tag)
