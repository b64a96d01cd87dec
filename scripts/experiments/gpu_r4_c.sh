#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
timeout 300 python scripts/r4_attn_delta_timeline.py 128
