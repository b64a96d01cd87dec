#!/bin/bash
# per-shape GEMM microbenchmark (heuristic) with the stock library and with every variant; stream-K off unless $SK is set
cp pevit_amd/libpevit_hip.so /tmp/stock.so
run() { python - <<PY 2>&1 | grep -v amdgpu
import sys; sys.path.insert(0, 'scripts')
import bench_gemm as bg
bg.tune("gemm_streamk", ${SK:-0})
bg.shapes("b32")
PY
}
echo "== stock"; run
for v in pevit_amd/variants/libpevit_hip_*.so; do
  tag=$(basename $v .so); cp $v pevit_amd/libpevit_hip.so
  echo "== ${tag#libpevit_hip_}"; run
done
cp /tmp/stock.so pevit_amd/libpevit_hip.so
