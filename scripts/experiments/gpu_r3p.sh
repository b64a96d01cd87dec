#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
echo "== fp8-act report"; timeout 900 python scripts/r3_fp8_act_report.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r3_fp8_act.md | tail -20
echo "== adapter tests"; timeout 900 python -m pytest tests/test_gpu_ops2.py tests/test_gpu_tower.py -x -q -m gpu -k "adapter or compacter or ln_bwd" 2>&1 | tail -4
echo "== adapter/compacter"; for m in adapter compacter kadaptation; do timeout 300 python bench.py --method $m --steps 50 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$m', round(d['value']), round(d['ms_per_step'],3))"; done
echo "== L/14 x3"; for w in bf16 fp8 fp8-act bf16 fp8 fp8-act; do timeout 600 python bench.py --arch ViT-L/14 --batch 32 --weights $w --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$w', round(d['value'],1), round(d['ms_per_step'],3), round(d['roofline']['achieved'],1), round(d['roofline']['gemm_ms_per_step'],3), round(d['roofline']['whole_step']['frac'],4))"; done
