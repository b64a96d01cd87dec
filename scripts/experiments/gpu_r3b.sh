#!/bin/bash
# round 3, call B: in-step kernel table with the staggered kernel; out-proj on the k-split tile; XCD-aligned bands
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
echo "== gemm tests"; timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "gemm" 2>&1 | tail -3
echo "== kstats default"; KSTATS_LINES=24 bash scripts/gpu_kstats.sh r3b_default | cut -c1-170
echo "== kstats ksplit_mink=512"; KSTATS_LINES=14 bash scripts/gpu_kstats.sh r3b_mink --tune gemm_ksplit_mink=512 | cut -c1-170
echo "== in-step A/B"; bash scripts/gpu_ab.sh "gemm_band=-1" "gemm_band=0" "gemm_ksplit_mink=512" "gemm_band=-1" "gemm_band=0" "gemm_ksplit_mink=512" "gemm_stagger=0"
