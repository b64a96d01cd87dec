#!/bin/bash
# round 5, call y: what the attention kernels wait for at N = 257 (ViT-L/14) and N = 50: LDS activity / bank conflicts / MFMA busy / wait states
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
R=$PWD
for cfg in "l14:--arch ViT-L/14 --batch 32" "b32:"; do
  tag=${cfg%%:*}; args=${cfg#*:}
  O=$R/gpurun_out/pmc_attn_$tag; rm -rf $O; mkdir -p $O
  ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES -d $O -o p -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-harness $args > $O/log.txt 2>&1 )
  python scripts/pmc_counters.py $(find $O -name "*.db" | head -1) "attn" > $O/counters.md 2>&1
  echo "== $tag"; cut -c1-260 $O/counters.md
done
