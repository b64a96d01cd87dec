#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
J='import sys,json; d=json.loads(sys.stdin.readline()); r=d["roofline"]; print(sys.argv[1], round(d["value"],1), round(d["ms_per_step"],3), "gemm TF/s", round(r["achieved"],1), "gemm ms", round(r.get("gemm_ms_per_step",0),3))'
echo "== all gpu tests"; timeout 2400 python -m pytest tests -q -m gpu 2>&1 | tail -6
echo "== pmc passes"; bash scripts/run_pmc_passes.sh r03 2>&1 | tail -30 | cut -c1-1500
echo "== batch 64 line"; timeout 300 python bench.py --batch 64 --no-cpu-baseline 2>/dev/null | tee gpurun_out/r3_bench_b64.json | cut -c1-300
echo "== methods"
for m in kadaptation lora adapter compacter; do timeout 600 python bench.py --method $m --steps 50 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "$J" "$m"; done
echo "== archs"
for a in "ViT-L/14 32 bf16" "ViT-L/14 32 fp8" "ViT-L/14 32 fp8-act" "ViT-B/16 64 bf16"; do set -- $a; timeout 600 python bench.py --arch $1 --batch $2 --weights $3 --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "$J" "$1 b$2 $3"; done
echo "== smoke"; timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3
