#!/bin/bash
# round 5, call h: graph replay mismatch diagnosis; GPU suite with the post-MLP bf16 stream; Adapter / Compacter A/B
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
timeout 300 python scripts/r5_graph_debug.py tiny-128 2>&1 | grep -v amdgpu.ids | tail -12
timeout 300 python scripts/r5_graph_debug.py tiny-128 gemm_streamk=0 2>&1 | grep -v amdgpu.ids | tail -6
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|Error|error|FAILED|assert" | tail -12
bash scripts/gpu_ab.sh "--method adapter" "gstream_bf16=0" "gstream_bf16=1"
bash scripts/gpu_ab.sh "--method compacter" "gstream_bf16=0" "gstream_bf16=1" | tail -4
