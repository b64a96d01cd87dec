#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
J='import sys,json; d=json.loads(sys.stdin.readline()); r=d["roofline"]; print(sys.argv[1], round(d["value"],1), round(d["ms_per_step"],3), "gemm ms", round(r.get("gemm_ms_per_step",0),3))'
echo "== ops + fp8 tests"; timeout 1500 python -m pytest tests/test_gpu_ops.py tests/test_gpu_ops2.py tests/test_gpu_fp8.py -x -q -m gpu 2>&1 | tail -4
echo "== methods"
for m in kadaptation lora adapter compacter; do timeout 600 python bench.py --method $m --steps 50 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "$J" "$m"; done
echo "== archs"
for a in "ViT-L/14 32 bf16" "ViT-L/14 32 fp8" "ViT-L/14 32 fp8-act" "ViT-B/16 64 bf16" "ViT-B/32 64 bf16"; do set -- $a; timeout 600 python bench.py --arch $1 --batch $2 --weights $3 --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "$J" "$1 b$2 $3"; done
echo "== adapter kstats"; KSTATS_LINES=30 bash scripts/gpu_kstats.sh r3ah_adapter --method adapter | cut -c1-150
