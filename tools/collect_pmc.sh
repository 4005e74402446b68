#!/bin/bash
# Counter passes for the bench workload, one rocprofv3 run per pass (counters are not split
# over passes automatically).  Usage: tools/collect_pmc.sh OUTDIR [bench args...]
set -u
OUT=$1; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
run_pass() {
    name=$1; shift
    timeout -s KILL 200 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/$name" -- python3 "$ROOT/bench.py" $BENCH_ARGS > "$OUT/$name.log" 2>&1
    rc=$?
    echo "pass $name rc $rc"
    return $rc
}
BENCH_ARGS="${*:---hyps 1048576 --steps 1 --warmup 0 --no-cpu-baseline}"
run_pass flops SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES || exit 1
run_pass busy SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY || exit 1
run_pass fetch FETCH_SIZE || exit 1
run_pass write WRITE_SIZE || exit 1
