#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
timeout 900 python -m pytest tests/test_gpu_ops2.py tests/test_gpu_mirror.py -x -q -k "uint8 or device_feeder" 2>&1 | grep -E "^E|passed|failed|Error" | head -30
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "^E|passed|failed|FAILED" | head -20
timeout 300 python scripts/time_sweep_reuse.py 2>&1 | tail -3
