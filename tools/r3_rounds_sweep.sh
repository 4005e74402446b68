#!/bin/bash
# Round lengths of the shared launches (first round, growth factor) against the list length.
# Usage: tools/r3_rounds_sweep.sh OUTFILE
set -u
OUT=$1
: > "$OUT"
for n in 100 200 400 800; do
  for r in 128,5 128,8 256,5 256,8 128,12 512,4 ; do
    v=$(POSE_AMD_SEG_ROUNDS=$r timeout -k 10 200 python bench.py --workload dubrovnik --early-exit --queries $n --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['steps_computed'], d['roofline']['launches'])") || exit 1
    echo "queries $n rounds $r : $v" >> "$OUT"
  done
done
