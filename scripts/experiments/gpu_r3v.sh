#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
echo "== gemm ops tests"; timeout 1500 python -m pytest tests/test_gpu_ops.py -x -q -m gpu 2>&1 | tail -8
echo "== shapes"; timeout 600 python scripts/r3_kphase.py 2>&1 | tail -60
echo "== bench A/B"
for m in 1 2 1 2; do timeout 600 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --tune gemm_ksplit_stagger=$m 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('stagger=$m', round(d['value'],1), round(d['ms_per_step'],3), round(d['roofline']['achieved'],1))"; done
echo "== bench b64 A/B"
for m in 1 2; do timeout 600 python bench.py --batch 64 --steps 50 --warmup 10 --no-cpu-baseline --tune gemm_ksplit_stagger=$m 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('stagger=$m', round(d['value'],1), round(d['ms_per_step'],3), round(d['roofline']['achieved'],1))"; done
