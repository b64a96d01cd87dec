#!/bin/bash
# round 5, end of round: the driver's own sequence on a fresh box -- GPU tests, smoke(), the default bench line (timed)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
echo "(GPU suite: see the previous run)"
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
T0=$(date +%s); python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; echo "bench wall $(( $(date +%s) - T0 )) s"; tail -2 gpurun_out/bench_default.err
tail -1 gpurun_out/bench_default.json | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print({k: d[k] for k in ('metric','value','unit','n_gpus','steps','warmup','ms_per_step','higher_is_better','scaling','vs_baseline','dtype','data')})
print(d['config'])
r=d['roofline']; print({k: r[k] for k in ('bound','achieved','peak','unit','frac','traffic','whole_step_frac','mfma_floor_ms','hbm_floor_ms','whole_step_hbm_frac','traffic_all')})
print(d['cpu_baseline']); print(d.get('harness_images_per_sec',{}).get('bs128'))"
wc -l gpurun_out/bench_default.json
