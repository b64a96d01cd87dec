#!/bin/bash
# round 5: harness rates after the feeder skips the device-side wait on uploads the host has seen complete
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
python -m pytest tests -x -q -m gpu -k "mirror or harness or feeder or train_one" 2>&1 | grep -E "passed|failed|Error" | tail -3
for r in 1 2; do python bench.py --no-cpu-baseline 2>/dev/null | grep '^{"metric' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], json.dumps(d.get('harness_images_per_sec')))"; done
