#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
J='import sys,json; d=json.loads(sys.stdin.readline()); r=d["roofline"]; print(sys.argv[1], round(d["value"],1), round(d["ms_per_step"],3), "gemm ms", round(r.get("gemm_ms_per_step",0),3))'
echo "== few-row tests"; timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "few_row" 2>&1 | tail -15
echo "== gemm ops tests"; timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -m gpu 2>&1 | tail -4
echo "== bench A/B skinny"
for m in 0 1 0 1; do timeout 600 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --tune gemm_skinny=$m 2>/dev/null | python -c "$J" "skinny=$m"; done
for m in 0 1; do timeout 600 python bench.py --batch 64 --steps 50 --warmup 10 --no-cpu-baseline --tune gemm_skinny=$m 2>/dev/null | python -c "$J" "b64 skinny=$m"; done
echo "== kstats skinny"; KSTATS_LINES=44 bash scripts/gpu_kstats.sh r3ad | grep -E "skinny|streamk|gemm_kernel|total" | cut -c1-150
