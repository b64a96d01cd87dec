#!/bin/bash
# round 5, end of round: the GPU suite twice more (flakiness), smoke(), a 3000-step soak of the headline step with loss / error-word check
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
for r in 1 2; do timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | grep -E "passed|failed|FAILED" | tail -3; done
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
python bench.py --steps 3000 --warmup 20 --no-cpu-baseline --no-harness 2>/dev/null | grep '^{"metric' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('soak', d['value'], d['ms_per_step'], d['median_ms_per_step'], d['config']['final_loss'])"
