#!/bin/bash
# Runs bench.py against alternative builds of the library: tools/sweep_lib.sh tools/var_*.so
for lib in "$@"; do
  echo "== $lib"
  POSE_AMD_LIB=$PWD/$lib timeout -k 10 120 python -u bench.py --hyps 1048576 --steps 2 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['kernel_ms_per_step'])" || exit 1
done
