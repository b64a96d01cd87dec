# LDS padding sweep of the attention kernels, compiled and timed on the GPU box.
cd pevit_amd/csrc
for tp in 4 8 12 20; do for ldr in 72 80 88 104; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -Wno-cuda-compat -ffp-contract=fast -DATT_TPAD=$tp -DATT_LDR=$ldr -c attention.hip -o attention.o 2>/dev/null
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC capi.o gemm.o norm.o attention.o lowrank.o misc.o stem_head.o adapter.o -o ../libpevit_hip.so
  echo "TPAD $tp LDR $ldr: $(cd ../..; python scripts/bench_attn.py 2>&1 | grep -E 'attn fwd|phase<=3: ' | head -2 | tr '\n' ' ')"
done; done
