#!/bin/bash
# round 5, call v: is the GPU suite stable?  three runs back to back (multi-process tests: ports, spawn, bounded waits)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
for r in 1 2 3; do timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | grep -E "passed|failed|FAILED|Error" | tail -3; done
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
