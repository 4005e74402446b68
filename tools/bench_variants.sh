#!/bin/bash
# Times library variants (pose_estimation_amd/libpose_amd_<name>.so) on the headline benchmark.
# Usage: tools/bench_variants.sh OUTDIR name...
OUT=$1; shift
mkdir -p "$OUT"
for v in "$@"; do
  lib=$PWD/pose_estimation_amd/libpose_amd_$v.so
  [ "$v" = base ] && lib=$PWD/pose_estimation_amd/libpose_amd.so
  POSE_AMD_LIB=$lib timeout -k 10 200 python bench.py --steps 10 --warmup 2 --no-cpu-baseline > "$OUT/$v.json" 2> "$OUT/$v.err" || { echo "$v failed"; tail -3 "$OUT/$v.err"; continue; }
  python - "$OUT/$v.json" "$v" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print(sys.argv[2], d["value"], d["ms_per_step"], d["roofline"]["kernel_ms_per_step"], "best", d["config"]["best_inliers"])
PY
done
