#!/bin/bash
# round 5: what compile-time token / head counts would buy the attention and low-rank kernels (ViT-B/32: N = 50, H = 12), same box
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
bash scripts/gpu_variants_args.sh "attn_|lowrank_combo|per step" 2>&1 | grep -v "^W2026" | cut -c1-150
find gpurun_out -name "*.db" -delete
