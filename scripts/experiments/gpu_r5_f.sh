#!/bin/bash
# round 5, call f: allreduce_flat at world 8 on one device -- diagnosis; graph capture A/B; PMC passes
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
timeout 300 python scripts/r5_ar_debug.py 8 24 2>&1 | grep -v "^$" | tail -30
echo "---- coarse"; PEVIT_AR_COARSE=1 timeout 300 python scripts/r5_ar_debug.py 8 24 2>&1 | grep -v "^$" | tail -14
echo "---- graph"
for r in 1 2 3; do for g in "" "--graph"; do
  timeout 300 python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-harness $g 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$g', '%.0f img/s  median %.3f ms' % (d['value'], d['median_ms_per_step']), d['config']['step_launch'])"
done; done
for g in "" "--graph"; do timeout 300 python bench.py --batch 64 --steps 100 --warmup 20 --no-cpu-baseline --no-harness $g 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('b64 $g', '%.0f img/s  median %.3f ms' % (d['value'], d['median_ms_per_step']))"; done
bash scripts/experiments/gpu_r5_pmc.sh
