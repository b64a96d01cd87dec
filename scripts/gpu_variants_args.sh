#!/bin/bash
# like gpu_variants.sh, with bench arguments: bash scripts/gpu_variants_args.sh "<kernel grep>" <bench args...>
PAT=$1; shift
cp pevit_amd/libpevit_hip.so /tmp/stock.so
echo "== stock"; KSTATS_LINES=40 bash scripts/gpu_kstats.sh stock "$@" | grep -E "$PAT|total kernel|images"
for v in pevit_amd/variants/libpevit_hip_*.so; do
  tag=$(basename $v .so); tag=${tag#libpevit_hip_}
  cp $v pevit_amd/libpevit_hip.so
  echo "== $tag"; KSTATS_LINES=40 bash scripts/gpu_kstats.sh $tag "$@" | grep -E "$PAT|total kernel|images"
done
cp /tmp/stock.so pevit_amd/libpevit_hip.so
