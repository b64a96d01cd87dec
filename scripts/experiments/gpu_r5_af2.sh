#!/bin/bash
# round 5: post-MLP adapter kernels with E = 256 NV as a compile-time constant against the runtime E (same box), + their tests
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
python -m pytest tests -x -q -m gpu -k "adapter or compacter" 2>&1 | grep -E "passed|failed|Error|error" | tail -5
for m in adapter compacter; do
  echo "#### $m"; bash scripts/gpu_variants_args.sh "adapter_fwd_kernel|adapter_bwd_kernel|per step" --method $m 2>&1 | grep -v "^W2026" | cut -c1-150
done
find gpurun_out -name "*.db" -delete
