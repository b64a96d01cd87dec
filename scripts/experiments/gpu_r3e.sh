#!/bin/bash
# round 3, call E: branch-free fast-path epilogue of the staggered kernel
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
echo "== gemm tests"; timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_fp8.py -x -q -m gpu -k "gemm" 2>&1 | tail -3
echo "== per-shape (us)"; timeout 600 python scripts/r3_gemm_shapes.py 2>&1 | grep -A4 "ksp1 \[full\] round 1\|old \[full\] round 1"
echo "== kstats default"; KSTATS_LINES=12 bash scripts/gpu_kstats.sh r3e_default | cut -c1-150
echo "== kstats no touch"; KSTATS_LINES=12 bash scripts/gpu_kstats.sh r3e_notouch --tune gemm_ablate=32 | grep -E "gemm8|total kernel" | cut -c1-150
echo "== kstats ablate=1 (epilogue only)"; KSTATS_LINES=12 bash scripts/gpu_kstats.sh r3e_ab1 --tune gemm_ablate=1 | grep -E "gemm8|total kernel" | cut -c1-150
echo "== in-step A/B"; bash scripts/gpu_ab.sh "gemm_stagger=1" "gemm_stagger=0" "gemm_ablate=32" "gemm_stagger=1" "gemm_stagger=0" "gemm_ablate=32"
