#!/bin/bash
# round 5, call m: head-chain merges (ce + loss_mean as one workgroup, dfeat's bf16 copy from the BatchNorm backward): GPU suite, bench at bs 128 / 64
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|FAILED|assert" | tail -8
for a in "" "--batch 64"; do for r in 1 2; do timeout 300 python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-harness $a 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$a', '%.0f img/s  median %.3f ms' % (d['value'], d['median_ms_per_step']))"; done; done
