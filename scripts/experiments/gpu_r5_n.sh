#!/bin/bash
# round 5, call n: scalar-path L2 touch ahead of the operand requests (gemm_l2touch = k-tiles ahead) in gemm_kphase_kernel
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
timeout 300 python scripts/r5_variant_check.py gemm_l2touch=2 2>&1 | tail -1
bash scripts/gpu_ab.sh "" "gemm_l2touch=0" "gemm_l2touch=1" "gemm_l2touch=2" "gemm_l2touch=4"
KSTATS_LINES=8 bash scripts/gpu_kstats.sh r5n0 --tune gemm_l2touch=0 | grep -E "kphase|per step"
KSTATS_LINES=8 bash scripts/gpu_kstats.sh r5n2 --tune gemm_l2touch=2 | grep -E "kphase|per step"
