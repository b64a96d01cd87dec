#!/bin/bash
# round-4 baseline on a fresh box: GPU test suite, default bench line, kernel-trace summary
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r4base
( time timeout 1500 python -m pytest tests -m gpu -x -q ) > gpurun_out/r4base/pytest.log 2>&1
tail -3 gpurun_out/r4base/pytest.log
timeout 300 python bench.py > gpurun_out/r4base/bench.json 2> gpurun_out/r4base/bench.err
cut -c1-600 gpurun_out/r4base/bench.json
bash scripts/gpu_kstats.sh r4base
