#!/bin/bash
# fused delta + attention forward: op test, in-step A/B
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -k "attn_fwd_delta" 2>&1 | grep -E "^E|passed|failed" | head -20
for t in fused_attn_delta=0 fused_attn_delta=1 fused_attn_delta=0 fused_attn_delta=1; do
  timeout 300 python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-harness --tune $t 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.readline()); r=d['roofline']; h=r['hbm_kernels']
print('$t', round(d['value']), round(d['ms_per_step'],3), round(d['median_ms_per_step'],3), {k:round(v['avg_us'],1) for k,v in h.items()})"
done
