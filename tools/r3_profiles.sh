#!/bin/bash
# Round-3 evidence: kernel-trace stats of the default bench command (the driver's form) and of the
# other workloads' own lines, then the counter passes of the default bench (one rocprofv3 run per pass).
# Usage (on the GPU box, from the repository root): tools/r3_profiles.sh gpurun_out/r3prof
set -u
OUT=$1
ROOT=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p "$OUT"; OUT=$(cd "$OUT" && pwd)
cd /tmp && export TMPDIR=/tmp && cd "$ROOT"
trace() {
    name=$1; shift
    timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/$name" -o "$name" -- python3 bench.py "$@" > "$OUT/$name.log" 2> "$OUT/$name.err"
    rc=$?
    echo "trace $name rc $rc"
    return $rc
}
trace bench --steps 20 --warmup 5 || exit 1
trace dub_exit --workload dubrovnik --early-exit --steps 3 --warmup 1 --no-cpu-baseline || exit 1
trace dub_adv --workload dubrovnik --advanced --steps 3 --warmup 1 --no-cpu-baseline || exit 1
trace dub_fixed --workload dubrovnik --steps 3 --warmup 1 --no-cpu-baseline || exit 1
trace p3p --known-focal --matches 2000 --inlier-frac 0.3 --no-cpu-baseline || exit 1
trace large --matches 100000 --inlier-frac 0.01 --hyps 2097152 --steps 3 --warmup 1 --no-cpu-baseline || exit 1
trace knn --workload knn --steps 5 --warmup 1 --no-cpu-baseline || exit 1
tools/collect_pmc.sh "$OUT/pmc" --hyps 1048576 --steps 1 --warmup 0 --no-cpu-baseline --no-extras || exit 1
echo done
