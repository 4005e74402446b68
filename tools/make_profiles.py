#!/usr/bin/env python3
"""Turns the raw rocprofv3 output merged back under gpurun_out/ into the committed summaries:
   tools/make_profiles.py TAG STATS_DIR PMC_DIR HYPS_PER_LAUNCH
writes profiles/TAG_bench_kernel_stats.csv, profiles/TAG_bench_pmc_summary.csv,
profiles/TAG_bench_under_rocprof.json (if STATS_DIR.log holds the bench line) and
profiles/TAG_traffic.json (per launch; FETCH_SIZE in KB doubled as measured for gfx950)."""
import csv
import glob
import io
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, stats_dir, pmc_dir, hyps = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4])
out = os.path.join(ROOT, "profiles")
os.makedirs(out, exist_ok=True)
ks = glob.glob(os.path.join(stats_dir, "**", "*kernel_stats.csv"), recursive=True)[0]
open(os.path.join(out, tag + "_bench_kernel_stats.csv"), "w").write(open(ks).read())
log = stats_dir.rstrip("/") + ".log"
if os.path.exists(log):
    lines = [l for l in open(log) if l.startswith("{")]
    if lines:
        open(os.path.join(out, tag + "_bench_under_rocprof.json"), "w").write(lines[-1])
passes = [os.path.join(pmc_dir, p) for p in ("flops", "busy", "fetch", "write") if os.path.isdir(os.path.join(pmc_dir, p))]
txt = subprocess.check_output([sys.executable, os.path.join(ROOT, "tools", "pmc_summary.py")] + passes).decode()
open(os.path.join(out, tag + "_bench_pmc_summary.csv"), "w").write(txt)
rows = {r["kernel"]: r for r in csv.DictReader(io.StringIO(txt))}
traffic = {"_note": "HBM-side bytes per launch from separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (KB units) over bench.py "
                    "launches of %d hypotheses each; FETCH_SIZE doubled as measured for wide reads on gfx950. The access widths "
                    "here (8-byte per-lane stores, scalar loads) are outside the calibrated cases: upper bounds." % hyps,
           "hypotheses_per_launch": hyps}
sys.path.insert(0, ROOT)
import bench  # noqa: E402  (the hash of the device sources these counters were collected with; bench.py flags a record whose hash differs)
traffic["kernel_source_sha256"] = bench.kernel_source_hash()
per_hyp = {}
for k, r in rows.items():
    if "pose_amd" not in k:
        continue
    name = k.replace("pose_amd::", "").split("<")[0]
    nd = float(r["dispatches"])
    if "FETCH_SIZE" in r and "WRITE_SIZE" in r:
        f, w = float(r["FETCH_SIZE"]) * 1024 * 2 / nd, float(r["WRITE_SIZE"]) * 1024 / nd
        traffic[name] = {"fetch_bytes": f, "write_bytes": w, "hbm_bytes_per_launch": f + w}
    if "SQ_INSTS_VALU_FMA_F64" in r:
        lane_ops = (2 * float(r["SQ_INSTS_VALU_FMA_F64"]) + float(r["SQ_INSTS_VALU_MUL_F64"]) + float(r["SQ_INSTS_VALU_ADD_F64"])
                    + float(r["SQ_INSTS_VALU_TRANS_F64"])) * 64 / nd / hyps
        per_hyp[name] = {"executed_fp64_lane_flop_per_hypothesis": round(lane_ops, 1),
                         "valu_instructions_per_hypothesis": round(float(r["SQ_INSTS_VALU"]) / nd / hyps, 2)}
traffic["_per_hypothesis"] = per_hyp
json.dump(traffic, open(os.path.join(out, tag + "_traffic.json"), "w"), indent=1)
print(json.dumps(per_hyp, indent=1))
print({k: v for k, v in traffic.items() if not k.startswith("_") and isinstance(v, dict)})
