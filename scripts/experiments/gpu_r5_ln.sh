#!/bin/bash
# round 5: LayerNorm kernels templated on E / 256 against the generic (predicated, four-group) ones, same box
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
python -m pytest tests -x -q -m gpu -k "ln or norm or layernorm or fp8 or step or tower" 2>&1 | grep -E "passed|failed|Error|error" | tail -5
bash scripts/gpu_variants_args.sh "ln_fwd_kernel|ln_bwd_kernel|per step" 2>&1 | grep -v "^W2026" | cut -c1-150
bash scripts/gpu_variants_args.sh "ln_fwd_kernel|ln_bwd_kernel|per step" --arch ViT-L/14 --batch 32 2>&1 | grep -v "^W2026" | cut -c1-150
find gpurun_out -name "*.db" -delete
