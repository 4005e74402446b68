#!/bin/bash
# Every benchmark line of the round, one after the other (1 GPU), the list-length sweep of the query
# batch and the two-rank rehearsals.  Usage: tools/r3_bench_all.sh OUTDIR
set -u
OUT=$1
mkdir -p "$OUT"
run() { name=$1; shift; timeout -k 10 400 python bench.py "$@" > "$OUT/$name.json" 2> "$OUT/$name.err"; rc=$?; echo "$name rc=$rc"; head -c 300 "$OUT/$name.json"; echo; return $rc; }
run bench --steps 20 --warmup 5 || exit 1
run bench100 || exit 1
run p3p --known-focal --matches 2000 --inlier-frac 0.3 || exit 1
run large --matches 100000 --inlier-frac 0.01 --hyps 2097152 --steps 3 --warmup 1 || exit 1
run dub_fixed --workload dubrovnik --steps 10 --warmup 2 || exit 1
run dub_exit --workload dubrovnik --early-exit --steps 20 --warmup 5 || exit 1
run dub_adv --workload dubrovnik --advanced --steps 20 --warmup 5 || exit 1
run dub_exit_host_queries --workload dubrovnik --early-exit --host-queries --steps 10 --warmup 2 --no-cpu-baseline || exit 1
run knn --workload knn --steps 5 --warmup 1 || exit 1
run gloo2_single --gpus 2 --backend gloo --steps 5 --warmup 1 --no-cpu-baseline || exit 1
run gloo2_dub_weak --gpus 2 --backend gloo --workload dubrovnik --early-exit --steps 3 --warmup 1 --no-cpu-baseline || exit 1
run gloo2_dub_strong --gpus 2 --backend gloo --workload dubrovnik --early-exit --scaling strong --steps 3 --warmup 1 --no-cpu-baseline || exit 1
# the collectives of the N > 1 path with one rank: the only way RCCL's part can run on a one-GPU box
run rccl1_single --gpus 1 --rehearse-collectives --steps 20 --warmup 3 --no-cpu-baseline || exit 1
run rccl1_dub --gpus 1 --rehearse-collectives --workload dubrovnik --early-exit --steps 5 --warmup 2 --no-cpu-baseline || exit 1
tools/r3_sweep.sh "$OUT/sweep" || exit 1
