#!/bin/bash
# one PMC pass of the bench step: bash scripts/gpu_pmc.sh TAG "COUNTER1 COUNTER2 ..." [kernel substring]
export TMPDIR=/tmp
R=$PWD; TAG=$1; O=$R/gpurun_out/pmc_$TAG
rm -rf $O; mkdir -p $O
cd /tmp
timeout 600 rocprofv3 --kernel-trace --pmc $2 -d $O -o p -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/log.txt 2>&1
cd $R
python scripts/pmc_counters.py $(find $O -name "*.db" | head -1) "$3" > $O/counters.md 2>&1
head -30 $O/counters.md | cut -c1-220
