#!/bin/bash
# end-of-round evidence: GPU suite, PMC passes (traffic json incl. all kernels, MFMA busy, kernel stats, bench line), other configurations
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|FAILED" | tail -6
bash scripts/experiments/gpu_r5_pmc.sh
bash scripts/gpu_r5_other_configs.sh
