#!/bin/bash
# round 5: post-MLP adapter kernels on 16-row tiles (two workgroups per CU) against the 32-row build; cross-entropy with batched rows
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
python -m pytest tests -x -q -m gpu -k "adapter or compacter or head or ce_ or loss" 2>&1 | grep -E "passed|failed|Error|error" | tail -5
for m in adapter compacter; do
  echo "#### $m"; bash scripts/gpu_variants_args.sh "adapter_fwd_kernel|adapter_bwd_kernel|per step|ce_loss" --method $m 2>&1 | grep -v "^W2026" | cut -c1-150
done
find gpurun_out -name "*.db" -delete
