#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
timeout 600 python -m pytest tests/test_gpu_tower.py tests/test_gpu_emulation.py -x -q -k "fused_post_mlp or adapter or compacter" 2>&1 | grep -E "^E|passed|failed" | head -20
for m in adapter compacter; do for t in adapter_fused=0 adapter_fused=1; do
  timeout 300 python bench.py --method $m --steps 60 --warmup 15 --no-cpu-baseline --no-harness --tune $t 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.readline()); r=d['roofline']; h=r['hbm_kernels']
print('$m $t', round(d['value']), round(d['median_ms_per_step'],3), {k:round(v['avg_us'],1) for k,v in h.items() if 'adapter' in k})"
done; done
