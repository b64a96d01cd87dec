#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_fp8.py -q -k gemm 2>&1 | tail -2
timeout 900 python scripts/bench_gemm.py --arch b32 --configs=0,7,1,8,6 --ablate --square > gpurun_out/r2l_gemm.log 2>&1
grep -E "^----|sum per|square|8192" gpurun_out/r2l_gemm.log
bash scripts/gpu_ab.sh "gemm_cfg_longk=0" "gemm_cfg_longk=7 --tune gemm_cfg_shortk=8" "gemm_cfg_longk=0" "gemm_cfg_longk=7 --tune gemm_cfg_shortk=8" "gemm_cfg_longk=0 --tune gemm_big=0" "gemm_cfg_shortk=6" "gemm_cfg_longk=6 --tune gemm_cfg_shortk=6"
