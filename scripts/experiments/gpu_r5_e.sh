#!/bin/bash
# round 5, call e: allreduce_flat at world 2 / 4 / 8 on one device + error paths, then the PMC passes
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
timeout 900 python -m pytest tests/test_gpu_allreduce.py tests/test_gpu_dp.py -q 2>&1 | grep -E "passed|failed|Error|error|FAILED|assert" | tail -12
bash scripts/experiments/gpu_r5_pmc.sh
