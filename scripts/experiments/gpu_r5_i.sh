#!/bin/bash
# round 5, call i: graph replay after the memset -> kernel change; ViT-L/14 bs 32 per-kernel tables, bf16 and fp8 weights (config 5)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
timeout 300 python scripts/r5_graph_debug.py tiny-128 2>&1 | grep -v amdgpu.ids | tail -7
timeout 300 python scripts/r5_graph_debug.py ViT-B/32 2>&1 | grep -v amdgpu.ids | tail -6
timeout 900 python -m pytest tests/test_gpu_mirror.py tests/test_gpu_tower.py tests/test_gpu_ops.py -q 2>&1 | grep -E "passed|failed|Error|error|FAILED|assert" | tail -6
KSTATS_LINES=22 bash scripts/gpu_kstats.sh l14bf16 --arch ViT-L/14 --batch 32
KSTATS_LINES=22 bash scripts/gpu_kstats.sh l14fp8 --arch ViT-L/14 --batch 32 --weights fp8
