#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
timeout 600 python -m pytest tests/test_gpu_tower.py -x -q -k "combined_lowrank" 2>&1 | grep -E "^E|passed|failed" | head -20
for t in lowrank_combo=0 lowrank_combo=1 lowrank_combo=0 lowrank_combo=1; do
  timeout 300 python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-harness --tune $t 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.readline()); r=d['roofline']; h=r['hbm_kernels']
print('$t', round(d['value']), round(d['median_ms_per_step'],3), {k:round(v['avg_us'],1) for k,v in h.items() if 'lowrank' in k})"
done
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "^E|passed|failed|FAILED" | head -10
