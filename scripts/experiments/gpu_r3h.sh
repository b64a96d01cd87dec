#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
echo "== emulation diag"; timeout 600 python scripts/r3_emul_diag.py 2>&1 | tail -12
echo "== gemm tests"; timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_fp8.py -x -q -m gpu -k "gemm" 2>&1 | tail -3
echo "== kstats default"; KSTATS_LINES=10 bash scripts/gpu_kstats.sh r3h_default | cut -c1-150
echo "== in-step A/B"; bash scripts/gpu_ab.sh "gemm_ksplit_stagger=1" "gemm_ksplit_stagger=0" "gemm_ksplit_stagger=1" "gemm_ksplit_stagger=0"
