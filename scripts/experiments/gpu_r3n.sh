#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
echo "== fp8 tests"; timeout 1500 python -m pytest tests/test_gpu_fp8.py -x -q -m gpu -k "fp8_x_fp8 or fp8_act" 2>&1 | tail -25
echo "== ViT-L/14 bs32: bf16 / fp8 / fp8-act"
for w in bf16 fp8 fp8-act bf16 fp8 fp8-act; do timeout 600 python bench.py --arch ViT-L/14 --batch 32 --weights $w --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$w', round(d['value'],1), round(d['ms_per_step'],3), round(d['roofline']['achieved'],1), round(d['roofline']['gemm_ms_per_step'],3))"; done
