#!/bin/bash
# round 5, call a: baseline of this round's box -- GPU tests, default bench line, kernel-trace table, vendor table (like for like)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r5a
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
timeout 600 python bench.py --steps 100 --warmup 20 > gpurun_out/r5a/bench_line.json 2> gpurun_out/r5a/bench_err.log; cut -c1-600 gpurun_out/r5a/bench_line.json
timeout 600 python scripts/r5_vendor_rotation.py 12 > gpurun_out/r5a/vendor_rotation.md 2> gpurun_out/r5a/vendor_err.log; cat gpurun_out/r5a/vendor_rotation.md; tail -3 gpurun_out/r5a/vendor_err.log
KSTATS_LINES=40 bash scripts/gpu_kstats.sh r5a
