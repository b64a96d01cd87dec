#!/bin/bash
# round 5: PMC passes (traffic json incl. the all-kernels figure, MFMA busy, kernel stats, bench line) on the current kernel sources
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
bash scripts/run_pmc_passes.sh r05 > gpurun_out/pmc_r05.log 2>&1
tail -2 gpurun_out/pmc_r05/traffic.txt; head -8 gpurun_out/pmc_r05/mfma_util.md | cut -c1-140; head -16 gpurun_out/pmc_r05/kernel_stats.md | cut -c1-130
python -c "
import json; d=json.loads(open('gpurun_out/pmc_r05/bench_line.json').read().strip().splitlines()[-1]); r=d['roofline']
print({k:d[k] for k in ('value','ms_per_step','median_ms_per_step')}); print({k:r.get(k) for k in ('frac','whole_step_frac','traffic','traffic_stale','traffic_all','traffic_all_over_algorithmic','hbm_floor_ms','mfma_floor_ms')}); print(d.get('harness_images_per_sec',{}).get('bs128'))"
grep -A8 all_kernels gpurun_out/hbm_traffic.json | head -12
