#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/r2d_tests.log 2>&1
tail -5 gpurun_out/r2d_tests.log
for big in 1 0 1 0; do
  timeout 300 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --tune gemm_big=$big 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('big=$big', round(d['value']), d['ms_per_step'], d['roofline']['achieved'], d['roofline']['gemm_ms_per_step'])"
done
for w in bf16 fp8 bf16 fp8; do
  timeout 600 python bench.py --arch ViT-L/14 --batch 32 --weights $w --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('L14 $w', round(d['value'],1), d['ms_per_step'], d['roofline']['achieved'], d['roofline']['gemm_ms_per_step'])"
done
