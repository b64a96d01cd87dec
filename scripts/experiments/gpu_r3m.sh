#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
echo "== fp8 + gemm tests"; timeout 1500 python -m pytest tests/test_gpu_fp8.py tests/test_gpu_ops.py -x -q -m gpu 2>&1 | tail -4
echo "== ViT-L/14 bs32: bf16 vs fp8 weights"
for w in bf16 fp8 bf16 fp8; do timeout 600 python bench.py --arch ViT-L/14 --batch 32 --weights $w --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$w', round(d['value'],1), round(d['ms_per_step'],3), round(d['roofline']['achieved'],1), round(d['roofline']['gemm_ms_per_step'],3))"; done
echo "== other configs"
timeout 600 python bench.py --arch ViT-B/32 --method lora --batch 128 --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('lora b32', round(d['value'],1), round(d['ms_per_step'],3))"
timeout 600 python bench.py --arch ViT-B/16 --method compacter --batch 64 --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('compacter b16', round(d['value'],1), round(d['ms_per_step'],3))"
timeout 600 python bench.py --arch ViT-B/32 --method adapter --batch 128 --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('adapter b32', round(d['value'],1), round(d['ms_per_step'],3))"
timeout 600 python bench.py --arch ViT-B/32 --method compacter --batch 128 --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('compacter b32', round(d['value'],1), round(d['ms_per_step'],3))"
