#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
echo "== block parity"; PARITY_VERBOSE=1 timeout 900 python scripts/r3_parity_table.py block 2>&1 | grep -v amdgpu.ids | tail -60
echo "== gpu tests (without the emulation file)"; timeout 1500 python -m pytest tests -x -q -m gpu --ignore=tests/test_gpu_emulation.py 2>&1 | tail -6
echo "== attention variants"; bash scripts/gpu_variants.sh "attn_bwd" 2>&1 | grep -v "^W2026" | cut -c1-150
