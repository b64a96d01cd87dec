#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
J='import sys,json; d=json.loads(sys.stdin.readline()); r=d["roofline"]; print(sys.argv[1], round(d["value"],1), round(d["ms_per_step"],3), "gemm ms", round(r.get("gemm_ms_per_step",0),3))'
echo "== tests"; timeout 1500 python -m pytest tests/test_gpu_ops.py tests/test_gpu_tower.py -x -q -m gpu 2>&1 | tail -3
echo "== bench"; for i in 1 2; do timeout 600 python bench.py --steps 50 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "$J" "default"; done
echo "== kstats"; KSTATS_LINES=60 bash scripts/gpu_kstats.sh r3an | grep -E "lowrank_reduce|total" | cut -c1-150
