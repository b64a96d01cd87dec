#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
echo "== compacter kstats"; KSTATS_LINES=48 bash scripts/gpu_kstats.sh r3at_compacter --method compacter | cut -c1-150
