#!/bin/bash
# ViT-L/14 B=32 (BASELINE config 5 shard): bf16 / fp8 / fp8-act lines + fp8 kernel table
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
for w in bf16 fp8 fp8-act bf16 fp8; do
  timeout 300 python bench.py --arch ViT-L/14 --batch 32 --weights $w --steps 40 --warmup 10 --no-cpu-baseline --no-harness $EXTRA 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.readline()); r=d['roofline']
print('$w', round(d['value'],1), 'img/s', round(d['median_ms_per_step'],3), 'ms  gemm', round(r['gemm_ms_per_step'],3), 'ms', r['launches_per_step'], 'launches  frac', round(r['frac'],3), 'whole', round(r['whole_step_frac'],3))"
done
KSTATS_LINES=30 bash scripts/gpu_kstats.sh r4l14fp8 --arch ViT-L/14 --batch 32 --weights fp8 | cut -c1-150
