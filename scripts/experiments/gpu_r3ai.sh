#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
echo "== all gpu tests"; timeout 2400 python -m pytest tests -q -m gpu 2>&1 | tail -8
