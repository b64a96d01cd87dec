set -x
export TMPDIR=/tmp
R=$PWD
mkdir -p $R/gpurun_out/pmc_fetch $R/gpurun_out/pmc_write $R/gpurun_out/ktrace
cd /tmp
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/pmc_fetch -o f -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --pmc-calib > $R/gpurun_out/pmc_fetch.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/gpurun_out/pmc_write -o w -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --pmc-calib > $R/gpurun_out/pmc_write.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/ktrace -o k -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $R/gpurun_out/ktrace.log 2>&1
cd $R
python bench.py --no-cpu-baseline > gpurun_out/bench_now.log 2>&1
tail -2 gpurun_out/pmc_fetch.log | cut -c1-600; tail -1 gpurun_out/bench_now.log | cut -c1-1500
find gpurun_out/pmc_fetch gpurun_out/pmc_write gpurun_out/ktrace -name "*.db" | head
