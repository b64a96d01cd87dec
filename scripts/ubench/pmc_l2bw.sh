export TMPDIR=/tmp
R=$PWD
cd /tmp
i=0
for set in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_TAG_STALL_sum" "TCC_EA0_RDREQ_sum TCC_BUSY_sum TCC_EA0_RDREQ_LEVEL_sum" "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum" "TCP_TCC_READ_REQ_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TOTAL_ACCESSES_sum" "TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_32B_sum TCC_LATENCY_FIFO_FULL_sum TCC_SRC_FIFO_FULL_sum"; do
  i=$((i+1))
  timeout 120 rocprofv3 --kernel-trace --pmc $set -d $R/gpurun_out/l2pmc$i -o p -- $R/scripts/ubench/l2bw pmc > $R/gpurun_out/l2pmc$i.log 2>&1 || echo "pass $i failed"
  tail -4 $R/gpurun_out/l2pmc$i.log | cut -c1-200
done
