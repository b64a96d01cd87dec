#!/bin/bash
# round 5: compile-time token count in the large-N attention kernels (ViT-B/16: 197, ViT-L/14: 257), same box; + tests of the
# templated attn_fwd_delta / lowrank_combo instances
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
python -m pytest tests -x -q -m gpu -k "lowrank or attn or attention or delta or tower or step" 2>&1 | grep -E "passed|failed|Error|error" | tail -5
cp pevit_amd/libpevit_hip.so /tmp/stock.so
run() { echo "== $1 $2"; KSTATS_LINES=40 bash scripts/gpu_kstats.sh $1 ${@:3} | grep -E "attn_|lowrank_combo|per step" | cut -c1-150; }
run stock b16 --arch ViT-B/16 --method compacter --batch 64
cp pevit_amd/variants/libpevit_hip_fix197.so pevit_amd/libpevit_hip.so; run fix197 b16 --arch ViT-B/16 --method compacter --batch 64
cp /tmp/stock.so pevit_amd/libpevit_hip.so; run stock l14 --arch ViT-L/14 --batch 32
cp pevit_amd/variants/libpevit_hip_fix257.so pevit_amd/libpevit_hip.so; run fix257 l14 --arch ViT-L/14 --batch 32
cp /tmp/stock.so pevit_amd/libpevit_hip.so; run stock b32
find gpurun_out -name "*.db" -delete
