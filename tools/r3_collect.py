#!/usr/bin/env python3
"""Copies what tools/r3_profiles.sh and tools/r3_bench_all.sh left under gpurun_out/ into the
committed summaries under profiles/:  tools/r3_collect.py [PROFDIR] [BENCHDIR]
(kernel-trace stats and the bench line printed under the profiler per workload, the counter
summaries of the default bench through tools/make_profiles.py -- with the hash of the device sources
they belong to -- the plain bench lines, and the list-length sweep as one table)."""
import glob
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
prof = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "r3prof")
bench = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "gpurun_out", "r3bench")
out = os.path.join(ROOT, "profiles")
if os.path.isdir(os.path.join(prof, "pmc")):
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "make_profiles.py"), "r3", os.path.join(prof, "bench"),
                           os.path.join(prof, "pmc"), "1048576"])
for name in ("dub_exit", "dub_adv", "dub_fixed", "p3p", "large", "knn"):
    ks = glob.glob(os.path.join(prof, name, "**", "*kernel_stats.csv"), recursive=True)
    if ks:
        open(os.path.join(out, "r3_%s_kernel_stats.csv" % name), "w").write(open(ks[0]).read())
    log = os.path.join(prof, name + ".log")
    lines = [l for l in open(log) if l.startswith("{")] if os.path.exists(log) else []
    if lines:
        open(os.path.join(out, "r3_%s_under_rocprof.json" % name), "w").write(lines[-1])
for name in ("bench", "bench100", "p3p", "large", "dub_fixed", "dub_exit", "dub_adv", "dub_exit_host_queries", "knn", "rccl1_single", "rccl1_dub", "gloo2_single", "gloo2_dub_weak",
             "gloo2_dub_strong"):
    f = os.path.join(bench, name + ".json")
    lines = [l for l in open(f) if l.startswith("{")] if os.path.exists(f) else []
    if lines:
        open(os.path.join(out, "r3_%s_line.json" % name), "w").write(lines[-1])
sweep = {}
for f in sorted(glob.glob(os.path.join(bench, "sweep", "*.json"))):
    lines = [l for l in open(f) if l.startswith("{")]
    if not lines:
        continue
    d = json.loads(lines[-1])
    kind, n = os.path.basename(f)[:-5].split("_q")
    sweep.setdefault(kind, {})[int(n)] = {"queries_per_s": d["value"], "ms_per_pass": d["ms_per_step"],
                                          "kernel_ms_per_pass": d["roofline"]["kernel_ms_per_step"], "steps_computed": d["roofline"].get("steps_computed")}
if sweep:
    for kind, t in sweep.items():
        if 800 in t and 100 in t:
            t["predicted_strong_scaling_1_to_8"] = round(t[800]["ms_per_pass"] / t[100]["ms_per_pass"], 2)
            t["predicted_strong_scaling_1_to_4"] = round(t[800]["ms_per_pass"] / t[200]["ms_per_pass"], 2) if 200 in t else None
            t["predicted_strong_scaling_1_to_2"] = round(t[800]["ms_per_pass"] / t[400]["ms_per_pass"], 2) if 400 in t else None
    sweep["_note"] = ("one GPU, resident lists of 100/200/400/800 queries (bench.py --workload dubrovnik --queries N): a GPU of an N-GPU strong-scaling run "
                      "over the 800-query list holds 800/N queries, so ms_per_pass(800)/ms_per_pass(800/N) is the speed-up such a run can reach; "
                      "exit = reference loop, adv = advanced processor, fixed = fixed budget of 10 000 hypotheses")
    json.dump(sweep, open(os.path.join(out, "r3_list_length_sweep.json"), "w"), indent=1)
print("profiles/ updated")
