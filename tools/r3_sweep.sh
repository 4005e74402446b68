#!/bin/bash
# List-length sweep on one GPU: a GPU of an N-GPU strong-scaling run of the 800-query batch holds
# 800/N queries, so queries/s at 100/200/400/800 queries is the 8/4/2/1-GPU curve in advance.
# Usage: tools/r3_sweep.sh OUTDIR
set -u
OUT=$1
mkdir -p "$OUT"
run() { name=$1; shift; timeout -k 10 300 python bench.py "$@" > "$OUT/$name.json" 2> "$OUT/$name.err"; rc=$?; echo "$name rc=$rc"; head -c 400 "$OUT/$name.json"; echo; return $rc; }
for n in 100 200 400 800; do
    run exit_q$n --workload dubrovnik --early-exit --queries $n --steps 20 --warmup 3 --no-cpu-baseline || exit 1
    run adv_q$n --workload dubrovnik --advanced --queries $n --steps 20 --warmup 3 --no-cpu-baseline || exit 1
    run fixed_q$n --workload dubrovnik --queries $n --steps 3 --warmup 1 --no-cpu-baseline || exit 1
done
