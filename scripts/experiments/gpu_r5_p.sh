#!/bin/bash
# round 5, call p: the driver's own sequence -- build check, smoke, the default bench command
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']
print({k:d[k] for k in ('value','ms_per_step','median_ms_per_step','n_gpus','steps','warmup','dtype','scaling','vs_baseline')})
print({k:r.get(k) for k in ('bound','frac','peak','unit','traffic','traffic_stale','mfma_floor_ms','hbm_floor_ms','whole_step_hbm_frac','traffic_all_over_algorithmic')})
print(d['cpu_baseline']); print(d['config'])"
