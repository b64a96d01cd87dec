#!/bin/bash
# round 5, call w: kernel table of BASELINE config 4's shard (ViT-B/16 + Compacter, 64 images)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
KSTATS_LINES=26 bash scripts/gpu_kstats.sh b16 --arch ViT-B/16 --method compacter --batch 64
