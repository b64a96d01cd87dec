#!/bin/bash
# round 5: both DP routes on a 1-rank RCCL group beside the fused step (bench.py --dp-route), batch 128 and 64
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
for b in 128 64; do
python bench.py --dp-route --batch $b --no-cpu-baseline --no-harness 2>/dev/null | grep '^{"metric' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(json.dumps({k: v for k, v in d['dp_route'].items() if k != 'how'})); print(d['value'], d['ms_per_step'])"
done
