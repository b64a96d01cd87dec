#!/bin/bash
# round 5, call z: unpadded swizzled LDS tiles in attention.hip (transposed reads 4-way -> 2-way conflicted): tests, kernel times on three towers
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_ops2.py tests/test_isa_hygiene.py -q 2>&1 | grep -E "passed|failed|FAILED|assert" | tail -5
KSTATS_LINES=14 bash scripts/gpu_kstats.sh z32 | grep -E "attn_|per step|images"
KSTATS_LINES=14 bash scripts/gpu_kstats.sh z16 --arch ViT-B/16 --method compacter --batch 64 | grep -E "attn_|per step|images"
KSTATS_LINES=14 bash scripts/gpu_kstats.sh z14 --arch ViT-L/14 --batch 32 | grep -E "attn_|per step|images"
find gpurun_out -name "*.db" -delete
