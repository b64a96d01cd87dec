#!/bin/bash
# round 5, call l: kernel tables at batch 64 and for the bottleneck Adapter
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
KSTATS_LINES=44 bash scripts/gpu_kstats.sh b64 --batch 64
KSTATS_LINES=30 bash scripts/gpu_kstats.sh adapter --method adapter
