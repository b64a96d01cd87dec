#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$PWD
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_fp8.py -x -q -k "gemm" 2>&1 | tail -2
for t in "gemm_cfg_longk=0" "gemm_cfg_longk=1" "gemm_cfg_longk=6" "gemm_cfg_shortk=6" "gemm_cfg_shortk=0" "gemm_cfg_longk=1 --tune gemm_cfg_shortk=6" "gemm_cfg_longk=6 --tune gemm_cfg_shortk=6" "gemm_cfg_longk=0"; do
  timeout 300 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --tune $t 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$t', round(d['value']), round(d['ms_per_step'],3), round(d['roofline']['achieved'],1), round(d['roofline']['gemm_ms_per_step'],3))"
done
