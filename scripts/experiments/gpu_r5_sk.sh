#!/bin/bash
# round 5: the K = 768 few-row products of the pruned last block on the few-row kernel (gemm_skinny_mink 24 -> 12), one slice (no
# hand-off) and three; bench A/B, alternating
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
for r in 1 2; do
for t in "gemm_skinny_mink=24" "gemm_skinny_mink=12" "gemm_skinny_mink=12 --tune gemm_skinny_slices=3"; do
  echo "== $t"; python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-harness --tune $t 2>/dev/null | grep '^{"metric' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['median_ms_per_step'])"
done; done
