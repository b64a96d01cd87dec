#!/bin/bash
# in-step A/B of tuning knobs: bash scripts/gpu_ab.sh "knob=val" "knob=val --tune knob2=val" ...
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
for t in "$@"; do
  timeout 300 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-harness --tune $t 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$t', round(d['value']), round(d['ms_per_step'],3), round(d['roofline']['achieved'],1), round(d['roofline']['gemm_ms_per_step'],3))"
done
