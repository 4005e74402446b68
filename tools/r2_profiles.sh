#!/bin/bash
# Round-2 evidence: kernel-trace stats of the default bench command and of the three Dubrovnik
# lines, then the counter passes of the default bench (one rocprofv3 run per pass).
# Usage (on the GPU box, from the repository root): tools/r2_profiles.sh gpurun_out/r2prof
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
trace bench --no-cpu-baseline || exit 1
trace dub_exit --workload dubrovnik --early-exit --steps 3 --warmup 1 --no-cpu-baseline || exit 1
trace dub_adv --workload dubrovnik --advanced --steps 3 --warmup 1 --no-cpu-baseline || exit 1
trace dub_fixed --workload dubrovnik --steps 3 --warmup 1 --no-cpu-baseline || exit 1
trace p3p --known-focal --matches 2000 --inlier-frac 0.3 --no-cpu-baseline || exit 1
trace large --matches 100000 --inlier-frac 0.01 --hyps 262144 --steps 5 --warmup 1 --no-cpu-baseline || exit 1
trace knn --workload knn --steps 5 --warmup 1 --no-cpu-baseline || exit 1
tools/collect_pmc.sh "$OUT/pmc" || exit 1
echo done
