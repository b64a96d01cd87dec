#!/bin/bash
# round 5: same-box A/B of the attention tile layouts (stock = unpadded swizzled; pad80 = rounds 1-4: 80-element rows, no swizzle) on three towers
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
for a in "" "--arch ViT-B/16 --method compacter --batch 64" "--arch ViT-L/14 --batch 32"; do
  echo "#### $a"; bash scripts/gpu_variants_args.sh "attn_fwd_kernel|attn_bwd_kernel|per step" $a 2>&1 | grep -v "^W2026" | cut -c1-150
done
find gpurun_out -name "*.db" -delete
