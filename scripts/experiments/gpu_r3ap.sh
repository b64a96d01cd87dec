#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
for w in bf16 fp8; do timeout 600 python bench.py --arch ViT-L/14 --batch 32 --weights $w --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$w', round(d['value'],1), round(d['ms_per_step'],3), round(d['roofline']['achieved'],1), round(d['roofline']['gemm_ms_per_step'],3)); [print('   ', k['epilogue'], k['M'], k['N'], k['K'], k['launches_per_step'], round(k['avg_us'],1), round(k['frac'],3)) for k in d['roofline']['per_kernel']]"; done
echo "== kstats fp8"; KSTATS_LINES=16 bash scripts/gpu_kstats.sh r3ap_fp8 --arch ViT-L/14 --batch 32 --weights fp8 | cut -c1-150
echo "== kstats bf16"; KSTATS_LINES=16 bash scripts/gpu_kstats.sh r3ap_bf16 --arch ViT-L/14 --batch 32 --weights bf16 | cut -c1-150
