#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r4f
( time timeout 900 python bench.py --no-cpu-baseline ) > gpurun_out/r4f/bench.json 2> gpurun_out/r4f/bench.err
tail -5 gpurun_out/r4f/bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r4f/bench.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step','median_ms_per_step')})
print(json.dumps(d.get('harness_images_per_sec'), indent=1))
PY
timeout 300 python scripts/time_sweep_reuse.py 2>&1 | tail -2
