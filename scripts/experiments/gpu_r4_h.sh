#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
python -m pytest tests/test_gpu_fp8.py -x -q 2>&1 | grep -E "^E|passed|failed" | head
for t in fp8_tail=0 fp8_tail=1 fp8_tail=0 fp8_tail=1; do
  timeout 300 python bench.py --arch ViT-L/14 --batch 32 --weights fp8 --steps 40 --warmup 10 --no-cpu-baseline --no-harness --tune $t 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.readline()); r=d['roofline']
print('$t', round(d['value'],1), 'img/s', round(d['median_ms_per_step'],3), 'ms', r['launches_per_step'], 'launches')"
done
for w in bf16 fp8-act; do
timeout 300 python bench.py --arch ViT-L/14 --batch 32 --weights $w --steps 40 --warmup 10 --no-cpu-baseline --no-harness 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.readline()); print('$w', round(d['value'],1), round(d['median_ms_per_step'],3))"
done
