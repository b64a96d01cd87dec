#!/bin/bash
# round 2, first GPU pass: parity of the tile configurations, then per-shape timings with ablations
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -k "gemm" > gpurun_out/r2a_tests.log 2>&1
tail -5 gpurun_out/r2a_tests.log
timeout 900 python scripts/bench_gemm.py --arch b32 --configs=-1,0,1,3,4,5 --ablate --square > gpurun_out/r2a_gemm_b32.log 2>&1
timeout 600 python scripts/bench_gemm.py --arch l14 --configs=0,3,4,5 > gpurun_out/r2a_gemm_l14.log 2>&1
tail -30 gpurun_out/r2a_gemm_b32.log
