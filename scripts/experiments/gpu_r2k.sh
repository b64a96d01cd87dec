#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
timeout 2300 python -m pytest tests -m gpu -q 2>&1 | grep -v amdgpu.ids > gpurun_out/r2k_tests.log; tail -4 gpurun_out/r2k_tests.log
for t in "gemm_cfg_longk=0" "gemm_cfg_longk=3" "gemm_cfg_longk=4" "gemm_cfg_shortk=3" "gemm_cfg_longk=0"; do
  timeout 300 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --tune $t 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$t', round(d['value']), round(d['ms_per_step'],3), round(d['roofline']['achieved'],1), round(d['roofline']['gemm_ms_per_step'],3))"
done
