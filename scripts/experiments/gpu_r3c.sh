#!/bin/bash
# round 3, call C: where the in-step GEMM time goes -- kernel tables with the epilogue stores ablated (gemm_ablate=2) and with the
# k-loop ablated (gemm_ablate=1); results of those runs are garbage, only the per-kernel durations are read
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
for ab in 0 2 1; do
  echo "== kstats gemm_ablate=$ab"; KSTATS_LINES=14 bash scripts/gpu_kstats.sh r3c_ab$ab --tune gemm_ablate=$ab | grep -E "gemm|total kernel" | cut -c1-150
done
echo "== in-step A/B"; bash scripts/gpu_ab.sh "side_stream=0" "side_stream=1" "gemm_ksplit_mink=512" "gemm_ksplit_mink=512 --tune side_stream=1"
