#!/bin/bash
# LDS counters of a bench workload: tools/pmc_lds.sh OUTDIR [bench args...]
set -u
OUT=$1; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS --output-format csv -d "$OUT/lds" -- python3 "$ROOT/bench.py" "$@" > "$OUT/lds.log" 2>&1
echo "rc $?"
