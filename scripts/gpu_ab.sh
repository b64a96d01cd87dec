#!/bin/bash
# same-box A/B of tune keys on the bench step: bash scripts/gpu_ab.sh "<bench args>" "key=a" "key=b" ...   (three alternating rounds)
ARGS=$1; shift
for round in 1 2 3; do
  for t in "$@"; do
    tune=""; for kv in $t; do tune="$tune --tune $kv"; done
    v=$(timeout 300 python bench.py --steps 60 --warmup 15 --no-cpu-baseline --no-harness $ARGS $tune 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.0f img/s  median %.3f ms' % (d['value'], d['median_ms_per_step']))")
    echo "round $round  [$ARGS] $t : $v"
  done
done
