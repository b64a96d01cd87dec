#!/bin/bash
# round 5, call c: bf16 gradient stream (gstream_bf16) -- GPU tests, in-step A/B, kernel table
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|Error|error" | tail -5
bash scripts/gpu_ab.sh "" "gstream_bf16=0" "gstream_bf16=1"
bash scripts/gpu_ab.sh "--batch 64" "gstream_bf16=0" "gstream_bf16=1"
bash scripts/gpu_ab.sh "--method lora" "gstream_bf16=0" "gstream_bf16=1" | tail -2
KSTATS_LINES=14 bash scripts/gpu_kstats.sh r5c
