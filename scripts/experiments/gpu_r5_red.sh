#!/bin/bash
# round 5: lowrank_reduce_kernel with 16-byte requests and the bias columns' chunks requested eight at a time
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
python -m pytest tests -x -q -m gpu -k "lowrank or tower or step or refinit or golden or fixture" 2>&1 | grep -E "passed|failed|Error" | tail -3
KSTATS_LINES=60 bash scripts/gpu_kstats.sh red | grep -E "lowrank_reduce|per step|images" | cut -c1-150
KSTATS_LINES=60 bash scripts/gpu_kstats.sh red2 --method lora | grep -E "lowrank_reduce|per step|images" | cut -c1-150
find gpurun_out -name "*.db" -delete
