#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
echo "== fp8 tests + streamk shared"; timeout 1500 python -m pytest tests/test_gpu_fp8.py tests/test_gpu_ops.py -x -q -m gpu -k "fp8_x_fp8 or fp8_act or shared" 2>&1 | tail -5
echo "== ViT-L/14 bs32: bf16 / fp8-act"
for w in bf16 fp8-act; do timeout 600 python bench.py --arch ViT-L/14 --batch 32 --weights $w --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$w', round(d['value'],1), round(d['ms_per_step'],3), round(d['roofline']['achieved'],1), round(d['roofline']['gemm_ms_per_step'],3)); [print('   ', k['epilogue'], k['M'], k['N'], k['K'], round(k['avg_us'],1), round(k['frac'],3)) for k in d['roofline']['per_kernel']]"; done
echo "== adapter kstats"; KSTATS_LINES=30 bash scripts/gpu_kstats.sh r3o_adapter --method adapter | cut -c1-150
