#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
J='import sys,json; d=json.loads(sys.stdin.readline()); r=d["roofline"]; print(sys.argv[1], round(d["value"],1), round(d["ms_per_step"],3), "gemm ms", round(r.get("gemm_ms_per_step",0),3))'
echo "== cfg tests with stagger=2"; PEVIT_X=1 timeout 900 python - <<'PY'
import ctypes as C, torch, sys
sys.path.insert(0, 'tests')
from pevit_amd import _lib
lib = _lib.load()
import test_gpu_ops as T
assert lib.pevit_tune(None, b"gemm_stagger", 2) == 0
for (M,N,K) in [(6400, 768, 768), (700, 2368, 256), (257, 136, 64), (1300, 640, 1024), (3200, 2368, 768)]:
    T.test_gemm_every_tile_config_is_bit_identical(lib, 3, M, N, K)
    print("ok", M, N, K, lib.pevit_debug_last_gemm_path())
PY
echo "== b64 A/B"
for m in 1 2 1 2; do timeout 600 python bench.py --batch 64 --steps 50 --warmup 10 --no-cpu-baseline --tune gemm_stagger=$m 2>/dev/null | python -c "$J" "b64 stagger=$m"; done
for a in "ViT-L/14 32" "ViT-B/16 64"; do set -- $a; for m in 1 2; do timeout 600 python bench.py --arch $1 --batch $2 --steps 30 --warmup 5 --no-cpu-baseline --tune gemm_stagger=$m 2>/dev/null | python -c "$J" "$1 b$2 stagger=$m"; done; done
