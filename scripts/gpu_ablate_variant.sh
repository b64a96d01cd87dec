#!/bin/bash
# stream-only ablation (gemm_ablate 10) with the stock library and with a variant: bash scripts/gpu_ablate_variant.sh TAG
cp pevit_amd/libpevit_hip.so /tmp/stock.so
python - <<'PY'
import sys; sys.path.insert(0, 'scripts')
import bench_gemm as bg
for cfg in (0, 1, 4, 5):
    bg.tune("gemm_config", cfg); bg.tune("gemm_ablate", 10)
    print("---- stock config", cfg, bg.NAMES[cfg], "[stream only]")
    bg.run("c_fc fwd", bg.EPI["BIAS_GELU"], 6400, 3072, 768, 768, 12, 50, iters=20)
    bg.run("c_proj fwd", bg.EPI["BIAS_RESID"], 6400, 768, 3072, 768, 12, 50, iters=20)
    bg.run("square 4096", bg.EPI["BF16"], 4096, 4096, 4096, 768, 12, 50, iters=10)
PY
cp pevit_amd/variants/libpevit_hip_$1.so pevit_amd/libpevit_hip.so
python - <<'PY'
import sys; sys.path.insert(0, 'scripts')
import bench_gemm as bg
for cfg in (0, 1, 4, 5):
    bg.tune("gemm_config", cfg); bg.tune("gemm_ablate", 10)
    print("---- variant config", cfg, bg.NAMES[cfg], "[stream only, free-running]")
    bg.run("c_fc fwd", bg.EPI["BIAS_GELU"], 6400, 3072, 768, 768, 12, 50, iters=20)
    bg.run("c_proj fwd", bg.EPI["BIAS_RESID"], 6400, 768, 3072, 768, 12, 50, iters=20)
    bg.run("square 4096", bg.EPI["BF16"], 4096, 4096, 4096, 768, 12, 50, iters=10)
PY
cp /tmp/stock.so pevit_amd/libpevit_hip.so
