#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
echo "== gemm ops tests (nl=2)"; PEVIT_TUNE=gemm_kphase_nl=2 timeout 1500 python -m pytest tests/test_gpu_ops.py -x -q -m gpu 2>&1 | tail -4
echo "== shapes"; timeout 600 python scripts/r3_kphase.py 6400 2>&1 | grep -v amdgpu.ids
echo "== bench A/B"
for m in 8 2 0 8 2 0; do timeout 600 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --tune gemm_kphase_nl=$m 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('nl=$m', round(d['value'],1), round(d['ms_per_step'],3), round(d['roofline']['achieved'],1))"; done
