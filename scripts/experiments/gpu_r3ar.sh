#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
J='import sys,json; d=json.loads(sys.stdin.readline()); r=d["roofline"]; print(sys.argv[1], round(d["value"],1), round(d["ms_per_step"],3), "final loss", d["config"].get("final_loss"))'
timeout 900 python bench.py --steps 1000 --warmup 20 --no-cpu-baseline 2>/dev/null | python -c "$J" "1000 steps"
timeout 900 python bench.py --steps 50 --warmup 10 2>/dev/null | cut -c1-400
