#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
echo "== cpu thread sweep"; timeout 900 python bench.py --cpu-sweep 2>/dev/null | tee gpurun_out/r3_cpu_sweep.jsonl | cut -c1-200
echo "== pmc passes"; bash scripts/run_pmc_passes.sh r03 2>&1 | tail -40 | cut -c1-1200
echo "== batch 64 line"; timeout 300 python bench.py --batch 64 --no-cpu-baseline 2>/dev/null | tee gpurun_out/r3_bench_b64.json | cut -c1-300
