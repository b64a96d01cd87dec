#!/bin/bash
# round 5: which factor of the attention tile layout costs ViT-B/16's backward 12 % -- row stride or swizzle (same box, five builds)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
bash scripts/gpu_variants_args.sh "attn_fwd_kernel|attn_bwd_kernel|per step" --arch ViT-B/16 --method compacter --batch 64 2>&1 | grep -v "^W2026" | cut -c1-150
find gpurun_out -name "*.db" -delete
