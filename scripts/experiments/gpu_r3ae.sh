#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
J='import sys,json; d=json.loads(sys.stdin.readline()); r=d["roofline"]; print(sys.argv[1], round(d["value"],1), round(d["ms_per_step"],3), "gemm ms", round(r.get("gemm_ms_per_step",0),3))'
echo "== slices"
for sl in 0 2 3 6; do timeout 600 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --tune gemm_skinny_slices=$sl 2>/dev/null | python -c "$J" "slices=$sl"; done
echo "== mink"
for mk in 24 12 8 24 12 8; do timeout 600 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --tune gemm_skinny_mink=$mk 2>/dev/null | python -c "$J" "mink=$mk"; done
echo "== kstats mink=8"; KSTATS_LINES=44 bash scripts/gpu_kstats.sh r3ae --tune gemm_skinny_mink=8 | grep -E "skinny|streamk|gemm_kernel|total" | cut -c1-150
