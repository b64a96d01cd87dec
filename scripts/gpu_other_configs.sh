#!/bin/bash
# bench.py on every BASELINE configuration's per-GPU shard + batch 64 -> gpurun_out/<round>_other_configs.{jsonl,md}
# usage (GPU box): bash scripts/gpu_other_configs.sh [r06]
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=${1:-r06}; export R
O=gpurun_out/${R}_other_configs.jsonl; : > $O
run() { timeout 400 python bench.py --steps 60 --warmup 15 --no-cpu-baseline --no-harness "$@" 2>/dev/null | tail -1 >> $O; }
run
run --batch 64
run --method lora
run --method adapter
run --method compacter
run --arch ViT-B/16 --method compacter --batch 64
run --arch ViT-L/14 --batch 32 --weights bf16
run --arch ViT-L/14 --batch 32 --weights fp8
run --arch ViT-L/14 --batch 32 --weights fp8-act
python - <<'PY'
import json, os
R=os.environ['R']
rows=[json.loads(l) for l in open(f'gpurun_out/{R}_other_configs.jsonl') if l.strip().startswith('{')]
out=[f"# Other configurations, {R} (one MI355X, one box for the whole table; `scripts/gpu_other_configs.sh`: `bench.py --steps 60 --warmup 15`)","",
"| configuration | images/s | ms / step (median) | whole-step frac of MFMA peak | GEMM family frac (events) | GEMM launches / step | non-GEMM ms / step |","|---|---|---|---|---|---|---|"]
for d in rows:
    r=d['roofline']; w=d['config']['workload'].split(' fine-tune')[0].replace('CLIP ','')
    b=d['config']['global_batch']; wt=d['dtype']
    out.append(f"| {w}, B = {b}, {wt} | {d['value']:.0f} | {d['median_ms_per_step']:.3f} | {r['whole_step_frac']:.3f} | {r['frac']:.3f} | {r['launches_per_step']:.0f} | {r['non_gemm_ms_per_step']:.3f} |")
out+=["","HBM-bound kernels of the headline configuration (HIP events, `roofline.hbm_kernels`):","","| kernel | launches / step | us | algorithmic MB | TB/s | of 8 TB/s |","|---|---|---|---|---|---|"]
for k,v in rows[0]['roofline']['hbm_kernels'].items():
    out.append(f"| {k} | {v['launches_per_step']:.0f} | {v['avg_us']:.1f} | {v['algorithmic_bytes_per_launch']/1e6:.1f} | {v['achieved_TBps']:.2f} | {v['frac_of_hbm_peak']:.2f} |")
open(f'gpurun_out/{R}_other_configs.md','w').write("\n".join(out)+"\n")
print("\n".join(out[:14]))
PY
