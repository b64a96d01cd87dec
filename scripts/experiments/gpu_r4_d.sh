#!/bin/bash
# PMC refresh (traffic json, MFMA busy, kernel stats) + kernel stats of batch 64 and of the Adapter
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
bash scripts/run_pmc_passes.sh r04a > gpurun_out/pmc_r04a.log 2>&1
tail -3 gpurun_out/pmc_r04a/traffic.txt; head -30 gpurun_out/pmc_r04a/kernel_stats.md | cut -c1-140
KSTATS_LINES=45 bash scripts/gpu_kstats.sh r4b64 --batch 64 | cut -c1-140
KSTATS_LINES=40 bash scripts/gpu_kstats.sh r4adapter --method adapter | cut -c1-140
