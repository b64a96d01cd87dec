#!/bin/bash
# round 5: where the DP routes lose their 0.1 ms: kernel trace of bench.py --dp-route, idle time around the optimizer kernel
cd /tmp && export TMPDIR=/tmp
R="$GRAFT_REPO_ROOT"; O=$R/gpurun_out/dpgaps; rm -rf $O; mkdir -p $O
timeout 600 rocprofv3 --kernel-trace -d $O -o k -- python $R/bench.py --dp-route --steps 30 --warmup 5 --no-cpu-baseline --no-harness > $O/bench.log 2>&1
cd $R
python scripts/r5_dp_gaps.py $(find $O -name "*.db" | head -1)
find gpurun_out -name "*.db" -delete
