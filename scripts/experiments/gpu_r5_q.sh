#!/bin/bash
# round 5, call q: sign projections of the full-size random-adapter fixtures on the production path and in the f32 verification mode
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
timeout 900 python -m pytest tests/test_gpu_tower.py tests/test_gpu_verify.py -q -s -k "full_size" 2>&1 | grep -E "projection-estimated|passed|failed|FAILED|assert" | tail -20
