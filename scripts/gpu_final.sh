#!/bin/bash
# end-of-round evidence in ONE call: GPU suite, PMC passes (traffic json incl. all kernels, MFMA busy, kernel stats, bench line), other
# configurations.  usage (GPU box):  bash scripts/gpu_final.sh [r06]     then here:  bash scripts/collect_profiles.sh r06
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=${1:-r06}
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | grep -E "passed|failed|FAILED|rror" | tail -8
bash scripts/run_pmc_passes.sh $R > gpurun_out/pmc_$R.log 2>&1
tail -2 gpurun_out/pmc_$R/traffic.txt; head -8 gpurun_out/pmc_$R/mfma_util.md | cut -c1-140; head -16 gpurun_out/pmc_$R/kernel_stats.md | cut -c1-130
python -c "
import json; d=json.loads(open('gpurun_out/pmc_$R/bench_line.json').read().strip().splitlines()[-1]); r=d['roofline']
print({k:d[k] for k in ('value','ms_per_step','median_ms_per_step')}); print({k:r.get(k) for k in ('frac','whole_step_frac','traffic','traffic_stale','traffic_all','traffic_all_over_algorithmic','hbm_floor_ms','mfma_floor_ms')}); print(d.get('harness_images_per_sec',{}).get('bs128'))"
bash scripts/gpu_other_configs.sh $R
