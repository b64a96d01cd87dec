#!/bin/bash
# end-of-round evidence: PMC passes (traffic json, MFMA busy, kernel stats, bench line), other configurations
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
bash scripts/run_pmc_passes.sh r04 > gpurun_out/pmc_r04.log 2>&1
tail -2 gpurun_out/pmc_r04/traffic.txt; head -8 gpurun_out/pmc_r04/mfma_util.md | cut -c1-140; head -24 gpurun_out/pmc_r04/kernel_stats.md | cut -c1-130
python -c "
import json; d=json.loads(open('gpurun_out/pmc_r04/bench_line.json').read().strip().splitlines()[-1]); r=d['roofline']
print({k:d[k] for k in ('value','ms_per_step','median_ms_per_step')}); print({k:r[k] for k in ('frac','whole_step_frac','traffic','traffic_stale')}); print(d.get('harness_images_per_sec',{}).get('bs128'))"
bash scripts/gpu_r4_other_configs.sh
