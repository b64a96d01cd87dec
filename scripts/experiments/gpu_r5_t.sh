#!/bin/bash
# round 5, call t: ViT-L/14 attention with 17 query tiles: waves per workgroup of the forward (8 -> 9 / 12) and of the backward (16 -> 9)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
bash scripts/gpu_variants_args.sh "attn_fwd_kernel|attn_bwd_kernel|per step" --arch ViT-L/14 --batch 32 2>&1 | grep -v "^W2026" | cut -c1-150
