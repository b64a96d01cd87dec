#!/bin/bash
# new partial-walk / generation tests, the GPU suite, and the new bench line
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r4a
( time timeout 1500 python -m pytest tests -m gpu -x -q ) > gpurun_out/r4a/pytest.log 2>&1
tail -5 gpurun_out/r4a/pytest.log
timeout 600 python bench.py --no-cpu-baseline > gpurun_out/r4a/bench.json 2> gpurun_out/r4a/bench.err
tail -3 gpurun_out/r4a/bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r4a/bench.json').read().strip().splitlines()[-1])
r=d['roofline']
print({k:d[k] for k in ('value','ms_per_step','median_ms_per_step','value_at_median')})
print({k:r[k] for k in ('achieved','frac','whole_step_frac','executed_gemm_frac_of_peak','non_gemm_ms_per_step','traffic','traffic_stale')})
for k,v in r['hbm_kernels'].items(): print(k, {a:round(b,3) for a,b in v.items()})
for e in r['per_kernel']: print({a:(round(b,3) if isinstance(b,float) else b) for a,b in e.items()})
PY
