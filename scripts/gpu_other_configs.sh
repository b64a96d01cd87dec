#!/bin/bash
# bench.py on the other BASELINE configurations (per-GPU shard sizes) -> gpurun_out/r02_other_configs.jsonl
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out; : > gpurun_out/r02_other_configs.jsonl
run() { timeout 900 python bench.py --no-cpu-baseline "$@" 2>/dev/null | tail -1 >> gpurun_out/r02_other_configs.jsonl; }
run --steps 50 --warmup 10
run --steps 50 --warmup 10 --method lora
run --steps 50 --warmup 10 --method adapter
run --steps 50 --warmup 10 --method compacter
run --steps 50 --warmup 10 --batch 64
run --steps 30 --warmup 5 --arch ViT-B/16 --method compacter --batch 64
run --steps 20 --warmup 5 --arch ViT-L/14 --batch 32
run --steps 20 --warmup 5 --arch ViT-L/14 --batch 32 --weights fp8
python - <<'PY'
import json
for l in open('gpurun_out/r02_other_configs.jsonl'):
    d=json.loads(l); r=d['roofline']
    print(d['metric'][26:], '|', round(d['value'],1), 'img/s |', round(d['ms_per_step'],3), 'ms | GEMM', round(r['achieved'],1), 'TF', round(r['frac'],3), '| step frac', round(r['whole_step']['frac'],3))
PY
