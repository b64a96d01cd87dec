#!/bin/bash
# round 5, call r: soak -- 3000 timed steps of the headline step (stream-K, bf16 gradient stream, merged head), the flat all-reduce at world 8 x 200 rounds
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
timeout 900 python bench.py --steps 3000 --warmup 20 --no-cpu-baseline --no-harness 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('soak 3000 steps:', '%.0f img/s  median %.3f ms  final loss %.6f' % (d['value'], d['median_ms_per_step'], d['config']['final_loss']))"
timeout 600 python scripts/r5_ar_debug.py 8 200 2>&1 | grep -E "wrong of|fine-grained" | sort | uniq -c | tail -4
