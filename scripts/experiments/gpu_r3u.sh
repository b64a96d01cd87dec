#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
echo "== new tests"; timeout 1500 python -m pytest tests/test_gpu_mirror.py tests/test_gpu_tower.py -x -q -m gpu -k "walking or hand_over or seam" 2>&1 | tail -12
echo "== all gpu tests"; timeout 2400 python -m pytest tests -q -m gpu 2>&1 | tail -6
