#!/bin/bash
# round 5, call u: attention backward with the 17th tile of N = 257 shared by the waves: kernel tests, ViT-L/14 table and rate
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_ops2.py -q -k "attn or attention" 2>&1 | grep -E "passed|failed|FAILED|assert" | tail -5
timeout 900 python -m pytest tests/test_gpu_tower.py tests/test_gpu_refinit.py tests/test_gpu_fp8.py -q 2>&1 | grep -E "passed|failed|FAILED|assert" | tail -5
KSTATS_LINES=12 bash scripts/gpu_kstats.sh l14u --arch ViT-L/14 --batch 32 | grep -E "attn_|per step|images"
for r in 1 2; do timeout 300 python bench.py --arch ViT-L/14 --batch 32 --steps 40 --warmup 10 --no-cpu-baseline --no-harness 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('L/14 bs32 bf16', '%.0f img/s  median %.3f ms' % (d['value'], d['median_ms_per_step']))"; done
