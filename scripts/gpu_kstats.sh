#!/bin/bash
# kernel-trace summary of the default bench step (on the GPU box): bash scripts/gpu_kstats.sh [tag] [bench args...]
export TMPDIR=/tmp
R=$PWD
TAG=${1:-k}; shift
O=$R/gpurun_out/kstats_$TAG
rm -rf $O; mkdir -p $O
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O -o k -- python $R/bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-harness "$@" > $O/bench.log 2>&1
cd $R
K=$(find $O -name "*.db" | head -1)
python scripts/prof_summary.py $K 55 > $O/kernel_stats.md 2>&1
head -${KSTATS_LINES:-24} $O/kernel_stats.md | cut -c1-150
tail -1 $O/bench.log | cut -c1-400
