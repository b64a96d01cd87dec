#!/bin/bash
for round in 1 2; do
for e in "X=1" "HIP_FORCE_DEV_KERNARG=1" "HIP_FORCE_DEV_KERNARG=0" "HSA_ENABLE_INTERRUPT=0" "HIP_FORCE_DEV_KERNARG=1 HSA_ENABLE_INTERRUPT=0"; do
  v=$(env $e timeout 300 python bench.py --steps 60 --warmup 15 --no-cpu-baseline --no-harness 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.0f img/s  median %.3f ms' % (d['value'], d['median_ms_per_step']))")
  echo "round $round [$e] : $v"
done
done
