#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
J='import sys,json; d=json.loads(sys.stdin.readline()); r=d["roofline"]; print(sys.argv[1], round(d["value"],1), round(d["ms_per_step"],3))'
echo "== lowrank tests"; timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "lowrank or delta" 2>&1 | tail -3
echo "== bench A/B lowrank_xcd"
for m in 0 1 0 1 0 1; do timeout 600 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --tune lowrank_xcd=$m 2>/dev/null | python -c "$J" "xcd=$m"; done
echo "== kstats xcd=1"; KSTATS_LINES=14 bash scripts/gpu_kstats.sh r3aa_x1 --tune lowrank_xcd=1 | grep -E "lowrank|total"
echo "== kstats xcd=0"; KSTATS_LINES=14 bash scripts/gpu_kstats.sh r3aa_x0 --tune lowrank_xcd=0 | grep -E "lowrank|total"
