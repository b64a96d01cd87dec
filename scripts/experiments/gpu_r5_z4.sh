#!/bin/bash
# round 5: per-instance attention tile layout (swizzled everywhere except the N = 197 backward): tests + three towers
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
python -m pytest tests -x -q -m gpu -k "attn or attention or tower or step or refinit" 2>&1 | tail -3
for a in "" "--arch ViT-B/16 --method compacter --batch 64" "--arch ViT-L/14 --batch 32"; do
  echo "#### $a"; KSTATS_LINES=40 bash scripts/gpu_kstats.sh stock $a 2>&1 | grep -E "attn_fwd_kernel|attn_bwd_kernel|per step|images" | cut -c1-150
done
find gpurun_out -name "*.db" -delete
