#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
J='import sys,json; d=json.loads(sys.stdin.readline()); r=d["roofline"]; print(sys.argv[1], round(d["value"],1), round(d["ms_per_step"],3), "gemm ms", round(r.get("gemm_ms_per_step",0),3))'
echo "== attention tests"; timeout 900 python -m pytest tests/test_gpu_ops.py -q -m gpu -k "attention" 2>&1 | grep -E "passed|failed"
for a in "ViT-L/14 32 bf16" "ViT-B/16 64 bf16" "ViT-B/32 128 bf16"; do set -- $a; timeout 600 python bench.py --arch $1 --batch $2 --weights $3 --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "$J" "$1 b$2 $3"; done
echo "== kstats L/14"; KSTATS_LINES=16 bash scripts/gpu_kstats.sh r3aq_l14 --arch ViT-L/14 --batch 32 | grep -E "attn|total" | cut -c1-150
echo "== kstats B/16"; KSTATS_LINES=16 bash scripts/gpu_kstats.sh r3aq_b16 --arch ViT-B/16 --batch 64 | grep -E "attn|total" | cut -c1-150
