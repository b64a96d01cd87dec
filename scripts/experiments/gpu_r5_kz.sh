#!/bin/bash
# round 5: two workgroups per 160x128 tile, half of K each (gemm_kz2), on the batch-64 step: tests, kernel table with and without
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
python -m pytest tests -x -q -m gpu -k "gemm" 2>&1 | grep -E "passed|failed|Error|error|assert" | tail -8
for kz in 1 0; do
  echo "== gemm_kz2=$kz, batch 64"; KSTATS_LINES=40 bash scripts/gpu_kstats.sh kz$kz --batch 64 --tune gemm_kz2=$kz | grep -E "gemm|per step|images" | cut -c1-150
done
python bench.py --batch 64 --steps 60 --warmup 15 --no-cpu-baseline --no-harness --tune gemm_kz2=1 2>/dev/null | tail -1 | cut -c1-330
python bench.py --batch 64 --steps 60 --warmup 15 --no-cpu-baseline --no-harness --tune gemm_kz2=0 2>/dev/null | tail -1 | cut -c1-330
find gpurun_out -name "*.db" -delete
