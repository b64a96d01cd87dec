#!/bin/bash
# round 3, call A: staggered 8-wave GEMM kernel -- correctness, per-shape timing, in-step A/B
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
echo "== gemm tests"; timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "gemm" 2>&1 | tail -5
echo "== per-shape (us)"; timeout 600 python scripts/r3_gemm_shapes.py 2>&1 | tail -40
echo "== in-step A/B"; bash scripts/gpu_ab.sh "gemm_stagger=0" "gemm_stagger=1" "gemm_stagger=1 --tune gemm_ksp=2" "gemm_stagger=0" "gemm_stagger=1"
echo "== ablation variants (stagger ksp1, no stores): ABL4 = no operand stream (compute only), ABL8 = no ds_read/MFMA (stream only)"
cp pevit_amd/libpevit_hip.so /tmp/stock.so
for v in ABL4 ABL8; do
  cp pevit_amd/variants/libpevit_hip_$v.so pevit_amd/libpevit_hip.so
  echo "-- $v"; timeout 300 python - <<'P'
import os, sys
sys.path.insert(0, "scripts")
import bench_gemm as bg
bg.tune("gemm_stagger", 1); bg.tune("gemm_ablate", 2)
for name, epi, M, N, K in [("c_fc fwd (gelu)", "BIAS_GELU", 6400, 3072, 768), ("qkv fwd", "QKV", 6400, 2368, 768), ("square 4096", "BF16", 4096, 4096, 4096)]:
    bg.run(name, bg.EPI[epi], M, N, K, 768, 12, 50, iters=20)
P
done
cp /tmp/stock.so pevit_amd/libpevit_hip.so
