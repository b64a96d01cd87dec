#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
echo "== gpu tests"; timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -6
echo "== B=64"; for t in "gemm_ksplit_small=1" "gemm_ksplit_small=0" "gemm_stagger=0"; do timeout 300 python bench.py --steps 50 --warmup 10 --batch 64 --no-cpu-baseline --tune $t 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$t', round(d['value']), round(d['ms_per_step'],3), round(d['roofline']['achieved'],1), round(d['roofline']['gemm_ms_per_step'],3), round(d['roofline']['whole_step']['frac'],4))"; done
echo "== B=64 kstats"; KSTATS_LINES=22 bash scripts/gpu_kstats.sh r3k_b64 --batch 64 | cut -c1-150
echo "== B=128"; bash scripts/gpu_ab.sh "gemm_stagger=1" "gemm_stagger=1"
