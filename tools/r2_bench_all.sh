#!/bin/bash
# Every benchmark line of the round, one after the other (1 GPU), plus the 16-worker kernel trace.
# Usage: tools/r2_bench_all.sh OUTDIR
set -u
OUT=$1
mkdir -p "$OUT"
run() { name=$1; shift; timeout -k 10 400 python bench.py "$@" > "$OUT/$name.json" 2> "$OUT/$name.err"; echo "$name rc=$?"; tail -c 600 "$OUT/$name.json" | head -c 300; echo; }
run single || exit 1
run p3p --known-focal --matches 2000 --inlier-frac 0.3 || exit 1
run large --matches 100000 --inlier-frac 0.01 --hyps 262144 --steps 5 --warmup 1 || exit 1
run dub_fixed --workload dubrovnik --steps 10 --warmup 2 || exit 1
run dub_exit --workload dubrovnik --early-exit --steps 20 --warmup 5 || exit 1
run dub_adv --workload dubrovnik --advanced --steps 20 --warmup 5 || exit 1
run gloo2_single --gpus 2 --backend gloo --steps 5 --warmup 1 --no-cpu-baseline || exit 1
run gloo2_dub --gpus 2 --backend gloo --workload dubrovnik --early-exit --steps 2 --warmup 1 --no-cpu-baseline || exit 1
