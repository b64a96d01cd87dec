#!/bin/bash
# round 5, call k: attn_fwd_delta with runs of three heads, two workgroups per CU (attn_delta_variant=1)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
timeout 300 python scripts/r5_variant_check.py attn_delta_variant=1 2>&1 | tail -2
bash scripts/gpu_ab.sh "" "attn_delta_variant=0" "attn_delta_variant=1"
KSTATS_LINES=12 bash scripts/gpu_kstats.sh r5k --tune attn_delta_variant=1 | grep -E "attn_fwd_delta|per step"
