#!/bin/bash
# round 5: host time of the parts of the single-exchange DP step (does the collective call hold the host?)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
python scripts/r5_dp_hosttime.py 2>&1 | grep -v "^W2026\|amdgpu.ids" | grep "host ms\|device ms"
