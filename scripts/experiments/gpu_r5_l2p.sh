#!/bin/bash
# round 5: feasibility of a side-stream L2 prefetch of the next product's weights (scripts/r5_l2_prefetch.py)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
timeout 600 python scripts/r5_l2_prefetch.py 2>&1 | grep -v "^W2026" | tee gpurun_out/r5_l2_prefetch.md
