#!/usr/bin/env python3
"""Copies what tools/r2_profiles.sh and tools/r2_bench_all.sh left under gpurun_out/ into the
committed summaries under profiles/:  tools/r2_collect.py [PROFDIR] [BENCHDIR]
(kernel-trace stats and the bench line printed under the profiler per workload, the counter
summaries of the default bench through tools/make_profiles.py, and the plain bench lines)."""
import glob
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
prof = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "r2prof")
bench = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "gpurun_out", "r2bench")
out = os.path.join(ROOT, "profiles")
subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "make_profiles.py"), "r2", os.path.join(prof, "bench"),
                       os.path.join(prof, "pmc"), "1048576"])
for name in ("dub_exit", "dub_adv", "dub_fixed", "p3p", "large", "knn"):
    ks = glob.glob(os.path.join(prof, name, "**", "*kernel_stats.csv"), recursive=True)
    if ks:
        open(os.path.join(out, "r2_%s_kernel_stats.csv" % name), "w").write(open(ks[0]).read())
    log = os.path.join(prof, name + ".log")
    lines = [l for l in open(log) if l.startswith("{")] if os.path.exists(log) else []
    if lines:
        open(os.path.join(out, "r2_%s_under_rocprof.json" % name), "w").write(lines[-1])
for name, dst in (("single", "bench"), ("p3p", "p3p"), ("large", "large"), ("dub_fixed", "dub_fixed"), ("dub_exit", "dub_exit"),
                  ("dub_adv", "dub_adv"), ("gloo2_single", "gloo2_single"), ("gloo2_dub", "gloo2_dub")):
    f = os.path.join(bench, name + ".json")
    lines = [l for l in open(f) if l.startswith("{")] if os.path.exists(f) else []
    if lines:
        open(os.path.join(out, "r2_%s_line.json" % dst), "w").write(lines[-1])
print("profiles/ updated")
