#!/bin/bash
# round 5, call j: fp8 B fragments read per k-step (no scratch in the fp8 k-loops): fp8 tests, ViT-L/14 bs 32 bf16 vs fp8, kernel table
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
timeout 900 python -m pytest tests/test_gpu_fp8.py tests/test_gpu_mirror.py -q 2>&1 | grep -E "passed|failed|Error|error|FAILED|assert" | tail -6
for r in 1 2 3; do for w in bf16 fp8; do
  timeout 300 python bench.py --arch ViT-L/14 --batch 32 --weights $w --steps 40 --warmup 10 --no-cpu-baseline --no-harness 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('L/14 bs32 $w', '%.0f img/s  median %.3f ms' % (d['value'], d['median_ms_per_step']))"
done; done
KSTATS_LINES=12 bash scripts/gpu_kstats.sh l14fp8b --arch ViT-L/14 --batch 32 --weights fp8
