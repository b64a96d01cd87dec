#!/bin/bash
# on the GPU box: for the stock library and every pevit_amd/variants/*.so -- the step digest (bit-identity, scripts/r6_lib_digest.py)
# and three alternating rounds of the bench step.   usage: bash scripts/gpu_lib_ab.sh [bench args...]
cp pevit_amd/libpevit_hip.so /tmp/stock.so
libs="stock"; for v in pevit_amd/variants/libpevit_hip_*.so; do t=$(basename $v .so); libs="$libs ${t#libpevit_hip_}"; done
use() { if [ "$1" = stock ]; then cp /tmp/stock.so pevit_amd/libpevit_hip.so; else cp pevit_amd/variants/libpevit_hip_$1.so pevit_amd/libpevit_hip.so; fi; }
for l in $libs; do use $l; echo "$l: $(python scripts/r6_lib_digest.py 2>/dev/null | tail -1)"; done
for round in 1 2 3; do
  for l in $libs; do
    use $l
    v=$(timeout 300 python bench.py --steps 60 --warmup 15 --no-cpu-baseline --no-harness "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.0f img/s  median %.3f ms' % (d['value'], d['median_ms_per_step']))")
    echo "round $round  $l : $v"
  done
done
use stock
