#!/bin/bash
# in-step A/B of the GEMM issue order / tile choice, then a kernel trace of the default bench
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$PWD
mkdir -p gpurun_out
for t in "gemm_big=1" "gemm_big=1 --tune gemm_burst=1" "gemm_big=0" "gemm_big=0 --tune gemm_burst=1" "gemm_big=1" "gemm_big=1 --tune gemm_burst=1"; do
  timeout 300 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --tune $t 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$t', round(d['value']), round(d['ms_per_step'],3), round(d['roofline']['achieved'],1), round(d['roofline']['gemm_ms_per_step'],3))"
done
export TMPDIR=/tmp
rm -rf gpurun_out/ktrace_r2e; mkdir -p gpurun_out/ktrace_r2e
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/ktrace_r2e -o k -- python $R/bench.py --steps 30 --warmup 5 --no-cpu-baseline > $R/gpurun_out/ktrace_r2e.log 2>&1
cd $R
DB=$(find gpurun_out/ktrace_r2e -name "*.db" | head -1)
python scripts/prof_summary.py $DB 45 > gpurun_out/r2e_kernel_stats.md
head -45 gpurun_out/r2e_kernel_stats.md
