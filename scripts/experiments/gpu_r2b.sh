#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -k "gemm" > gpurun_out/r2b_tests.log 2>&1
tail -3 gpurun_out/r2b_tests.log
timeout 900 python scripts/bench_gemm.py --arch b32 --configs=-1,0,1,3,4,5 --ablate --square > gpurun_out/r2b_gemm_b32.log 2>&1
grep -E "^----|sum per|square|8192" gpurun_out/r2b_gemm_b32.log
timeout 1200 python -m pytest tests/test_gpu_fp8.py -x -q > gpurun_out/r2b_fp8.log 2>&1
tail -15 gpurun_out/r2b_fp8.log
