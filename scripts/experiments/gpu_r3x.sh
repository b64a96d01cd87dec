#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
echo "== kadaptation kstats"; KSTATS_LINES=44 bash scripts/gpu_kstats.sh r3x_kad | cut -c1-170
echo "== adapter kstats"; KSTATS_LINES=44 bash scripts/gpu_kstats.sh r3x_adapter --method adapter | cut -c1-170
echo "== b64 kstats"; KSTATS_LINES=30 bash scripts/gpu_kstats.sh r3x_b64 --batch 64 | cut -c1-170
