#!/bin/bash
# Build compile-time variants of the library for an in-step A/B:  bash scripts/build_variants.sh "tag:file.hip:-DX=1 -DY=2" ...
# -> pevit_amd/variants/libpevit_hip_<tag>.so (git-ignored, travels to the GPU box).  Run them with scripts/gpu_variants.sh.
set -e
cd "$(dirname "$0")/../pevit_amd/csrc"
make -s
mkdir -p ../variants
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -Wno-cuda-compat -ffp-contract=fast"
for spec in "$@"; do
  tag=${spec%%:*}; rest=${spec#*:}; file=${rest%%:*}; defs=${rest#*:}
  /opt/rocm/bin/hipcc $FLAGS $defs -c $file -o /tmp/variant_$tag.o
  objs=""
  for o in $(sed -n "s/^SRCS = //p" Makefile | sed "s/\.hip//g"); do
    if [ "$o.hip" == "$file" ]; then objs="$objs /tmp/variant_$tag.o"; else objs="$objs $o.o"; fi
  done
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs -o ../variants/libpevit_hip_$tag.so
  echo "built $tag ($file $defs)"
done
