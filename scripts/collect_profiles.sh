#!/bin/bash
# copy the summaries of the last scripts/gpu_final.sh call from gpurun_out/ (scratch) into profiles/ (tracked)
cd "$(dirname "$0")/.."
R=${1:-r06}
cp gpurun_out/hbm_traffic.json profiles/hbm_traffic.json
cp gpurun_out/pmc_$R/kernel_stats.md profiles/${R}_bench_kernel_stats.md
tail -1 gpurun_out/pmc_$R/bench_line.json > profiles/${R}_bench_line.json
cp gpurun_out/pmc_$R/mfma_util.md profiles/${R}_mfma_utilisation.md
cp gpurun_out/pmc_$R/traffic.txt profiles/${R}_pmc_hbm_traffic.txt
cp gpurun_out/${R}_other_configs.jsonl gpurun_out/${R}_other_configs.md profiles/ 2>/dev/null
cp gpurun_out/pmc_$R/step_timeline.md profiles/${R}_step_timeline.md 2>/dev/null
