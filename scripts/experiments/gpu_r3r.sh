#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
echo "== adapter/compacter tests"; timeout 1500 python -m pytest tests/test_gpu_tower.py tests/test_gpu_emulation.py -x -q -m gpu -k "adapter or compacter" 2>&1 | tail -3
echo "== fused on/off"; for m in adapter compacter; do for f in 1 0 1 0; do timeout 300 python bench.py --method $m --steps 50 --warmup 10 --no-cpu-baseline --tune fused_bottleneck=$f 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$m fused=$f', round(d['value']), round(d['ms_per_step'],3))"; done; done
echo "== adapter kstats"; KSTATS_LINES=12 bash scripts/gpu_kstats.sh r3r_adapter --method adapter | grep -E "bottleneck|total" | cut -c1-150
