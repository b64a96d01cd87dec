#!/bin/bash
# round 3, call F: bias hoisted out of the k-split / 4-wave epilogues, out-proj on the k-split tile by default, no L2 touch
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
echo "== gemm tests"; timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_fp8.py -x -q -m gpu -k "gemm" 2>&1 | tail -3
echo "== kstats default"; KSTATS_LINES=26 bash scripts/gpu_kstats.sh r3f_default | cut -c1-150
echo "== in-step A/B"; bash scripts/gpu_ab.sh "gemm_ksplit_mink=512" "gemm_ksplit_mink=1024" "gemm_ksplit_mink=512" "gemm_ksplit_mink=1024"
echo "== full gpu tests"; timeout 1200 python -m pytest tests -x -q -m gpu 2>&1 | tail -5
