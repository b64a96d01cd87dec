#!/bin/bash
# round 5, call d: full GPU suite on the bf16 gradient stream (fp8 included), new bench line fields
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
timeout 1200 python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|Error|error|FAILED" | tail -12
timeout 600 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-harness > gpurun_out/r5d_bench.json 2> gpurun_out/r5d_bench.err; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r5d_bench.json').read().strip().splitlines()[-1]); r=d['roofline']
print({k:d[k] for k in ('value','ms_per_step','median_ms_per_step')})
print({k:r.get(k) for k in ('frac','whole_step_frac','mfma_floor_ms','hbm_floor_ms','whole_step_hbm_frac','whole_step_flop_per_byte','whole_step_bound','traffic_all','gemm_ms_per_step','non_gemm_ms_per_step')})
for k in r['per_kernel']: print({x:(round(k[x],3) if isinstance(k[x],float) else k[x]) for x in ('epilogue','N','K','avg_us','frac','bound','frac_of_hbm_peak','flop_per_byte')})
for n,k in r['hbm_kernels'].items(): print(n,{x:(round(k[x],3) if isinstance(k[x],float) else k[x]) for x in k})
PY
tail -3 gpurun_out/r5d_bench.err
bash scripts/gpu_ab.sh "--arch ViT-L/14 --batch 32 --weights fp8" "gstream_bf16=0" "gstream_bf16=1" | tail -4
