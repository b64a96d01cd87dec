#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
echo "== tower errors"; timeout 600 python scripts/r3_tower_errors.py 2>&1 | grep -v amdgpu.ids | tail -12
echo "== emulation tests"; PEVIT_RECORD_PARITY=$PWD/gpurun_out/parity_errors.jsonl timeout 1500 python -m pytest tests/test_gpu_emulation.py -q -m gpu 2>&1 | tail -8
echo "== bench"; timeout 600 python bench.py --steps 50 --warmup 10 2>/dev/null | tail -1 | cut -c1-600
