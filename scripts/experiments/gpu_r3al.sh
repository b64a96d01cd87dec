#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "lowrank" 2>&1 | tail -2
bash scripts/gpu_variants.sh "lowrank_u" 2>&1 | cut -c1-150
for v in pevit_amd/variants/libpevit_hip_*.so; do cp pevit_amd/libpevit_hip.so /tmp/stock.so; cp $v pevit_amd/libpevit_hip.so; echo "== test $v"; timeout 300 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "lowrank" 2>&1 | tail -1; cp /tmp/stock.so pevit_amd/libpevit_hip.so; done
