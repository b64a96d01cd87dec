#!/bin/bash
# round 5, call g: full GPU suite (graph replay, BN counter, allreduce world 2/4/8 + error paths), world-8 soak, PMC passes
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|Error|error|FAILED|assert" | tail -12
timeout 300 python scripts/r5_ar_debug.py 8 48 2>&1 | grep -v "^$" | grep -v amdgpu.ids | tail -12
bash scripts/experiments/gpu_r5_pmc.sh
