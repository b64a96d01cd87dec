#!/bin/bash
# round 5, call b: the *_refinit fixtures on the production kernels (stated gates)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
timeout 900 python -m pytest tests/test_gpu_refinit.py -q -s 2>&1 | grep -v "^$" | cut -c1-1500 | tail -40
