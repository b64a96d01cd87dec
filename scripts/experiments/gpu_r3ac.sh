#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
J='import sys,json; d=json.loads(sys.stdin.readline()); r=d["roofline"]; print(sys.argv[1], round(d["value"],1), round(d["ms_per_step"],3), "gemm ms", round(r.get("gemm_ms_per_step",0),3))'
for sh in 0 2 3 4 6 8 12; do
  if [ $sh == 0 ]; then T=""; else T="--tune gemm_streamk=2 --tune gemm_sk_share=$sh"; fi
  timeout 600 python bench.py --steps 50 --warmup 10 --no-cpu-baseline $T 2>/dev/null | python -c "$J" "share=$sh"
done
for sh in 3 6; do
echo "== kstats share=$sh"; KSTATS_LINES=40 bash scripts/gpu_kstats.sh r3ac_$sh --tune gemm_streamk=2 --tune gemm_sk_share=$sh | grep -E "streamk|gemm_kernel|total" | cut -c1-150
done
