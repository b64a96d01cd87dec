#!/bin/bash
# round 5, call x: the shared-tile scratch reserved only where used (ViT-B/16 keeps two attention-backward workgroups per CU), then the final evidence pass
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
KSTATS_LINES=12 bash scripts/gpu_kstats.sh b16b --arch ViT-B/16 --method compacter --batch 64 | grep -E "attn_bwd|per step|images"
bash scripts/gpu_r5_final.sh
