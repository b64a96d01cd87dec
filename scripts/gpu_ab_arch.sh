#!/bin/bash
# in-step A/B of tuning knobs on another configuration: bash scripts/gpu_ab_arch.sh "<bench args>" "knob=val" ...
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
ARGS=$1; shift
for t in "$@"; do
  timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline $ARGS --tune $t 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$ARGS', '$t', round(d['value'],1), round(d['ms_per_step'],3), round(d['roofline']['achieved'],1), round(d['roofline']['gemm_ms_per_step'],3))"
done
