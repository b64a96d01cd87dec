#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
J='import sys,json; d=json.loads(sys.stdin.readline()); r=d["roofline"]; print(sys.argv[1], round(d["value"],1), round(d["ms_per_step"],3), "gemm ms", round(r.get("gemm_ms_per_step",0),3))'
cp pevit_amd/libpevit_hip.so /tmp/stock.so
for rnd in 1 2; do
for v in stock nt1 nt2 nt3; do
  if [ $v == stock ]; then cp /tmp/stock.so pevit_amd/libpevit_hip.so; else cp pevit_amd/variants/libpevit_hip_$v.so pevit_amd/libpevit_hip.so; fi
  timeout 600 python bench.py --steps 50 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "$J" "$v"
done; done
cp /tmp/stock.so pevit_amd/libpevit_hip.so
bash scripts/gpu_variants.sh "gemm|ln_|attn|delta|lowrank" 2>&1 | cut -c1-150
