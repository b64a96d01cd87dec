#!/bin/bash
# rocprofv3 passes behind bench.py's roofline object: kernel trace + stats, HBM-side traffic (FETCH_SIZE and WRITE_SIZE in
# separate passes, each self-calibrated on a known 1 GiB stream), and matrix-core utilisation (SQ_VALU_MFMA_BUSY_CYCLES).
# PMC passes carry --kernel-trace only (gpurun refuses PMC combined with the API trace domains).
# usage (on the GPU box, from the repo root):  bash scripts/run_pmc_passes.sh [tag]
set -x
export TMPDIR=/tmp
R=$PWD
TAG=${1:-r06}
O=$R/gpurun_out/pmc_$TAG
rm -rf $O; mkdir -p $O/fetch $O/write $O/ktrace $O/mfma
cd /tmp
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/fetch -o f -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-harness --pmc-calib > $O/fetch.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/write -o w -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-harness --pmc-calib > $O/write.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d $O/mfma -o m -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-harness > $O/mfma.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats -d $O/ktrace -o k -- python $R/bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-harness > $O/ktrace.log 2>&1   # 30 timed + 5 warm-up + 2 x 10 event-profiled steps = 55
cd $R
F=$(find $O/fetch -name "*.db" | head -1); W=$(find $O/write -name "*.db" | head -1); K=$(find $O/ktrace -name "*.db" | head -1); M=$(find $O/mfma -name "*.db" | head -1)
python scripts/pmc_traffic.py $F $W "ViT-B/32|kadaptation|bs128" profiles/hbm_traffic.json > $O/traffic.txt 2>&1
cp profiles/hbm_traffic.json $R/gpurun_out/hbm_traffic.json     # profiles/ does not travel back from the GPU box: copy it from gpurun_out/
python scripts/prof_summary.py $K 55 > $O/kernel_stats.md 2>&1
python scripts/pmc_mfma.py $M > $O/mfma_util.md 2>&1
python scripts/r6_step_timeline.py $K > $O/step_timeline.md 2>&1
python bench.py --strict-traffic > $O/bench_line.json 2>$O/bench_err.log
tail -3 $O/traffic.txt; head -12 $O/mfma_util.md; head -8 $O/kernel_stats.md; cut -c1-900 $O/bench_line.json
