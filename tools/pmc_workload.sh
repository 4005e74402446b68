#!/bin/bash
# Two counter passes (instruction mix, busy / wait cycles) over any bench.py workload, one rocprofv3
# run per pass:  tools/pmc_workload.sh OUTDIR <bench.py args...>
set -u
OUT=$1; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p "$OUT"; OUT=$(cd "$OUT" && pwd)
cd /tmp && export TMPDIR=/tmp
run_pass() {
    name=$1; shift
    timeout -s KILL 200 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/$name" -- python3 "$ROOT/bench.py" $BENCH_ARGS > "$OUT/$name.log" 2>&1
    rc=$?
    echo "pass $name rc $rc"
    return $rc
}
BENCH_ARGS="$*"
run_pass mix SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_MFMA SQ_INSTS_SMEM || exit 1
run_pass busy SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY || exit 1
