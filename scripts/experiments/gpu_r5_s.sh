#!/bin/bash
# round 5, call s: allreduce_flat with the push as a kernel (system-scope release) -- tests, world-8 soak (kernel push, then the copy-engine push)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
timeout 900 python -m pytest tests/test_gpu_allreduce.py tests/test_gpu_dp.py -q 2>&1 | grep -E "passed|failed|FAILED|assert" | tail -6
( time timeout 900 python scripts/r5_ar_debug.py 8 400 ) 2>&1 | grep -E "wrong|real|never|different" | grep -v "0 wrong" | tail -12
echo "---- dma push"
( time PEVIT_AR_PUSH=dma timeout 900 python scripts/r5_ar_debug.py 8 200 ) 2>&1 | grep -E "wrong|real|never|different" | grep -v " 0 wrong" | tail -12
