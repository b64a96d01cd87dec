#!/bin/bash
# on the GPU box: kernel-trace summary of the bench step for the stock library and every pevit_amd/variants/*.so
# usage: bash scripts/gpu_variants.sh "<grep pattern of kernel names>"
PAT=${1:-kernel}
cp pevit_amd/libpevit_hip.so /tmp/stock.so
echo "== stock"; KSTATS_LINES=40 bash scripts/gpu_kstats.sh stock | grep -E "$PAT|total kernel|images"
for v in pevit_amd/variants/libpevit_hip_*.so; do
  tag=$(basename $v .so); tag=${tag#libpevit_hip_}
  cp $v pevit_amd/libpevit_hip.so
  echo "== $tag"; KSTATS_LINES=40 bash scripts/gpu_kstats.sh $tag | grep -E "$PAT|total kernel|images"
done
cp /tmp/stock.so pevit_amd/libpevit_hip.so
