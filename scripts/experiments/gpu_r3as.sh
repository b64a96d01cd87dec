#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 50 --warmup 10 --no-cpu-baseline 2>gpurun_out/r3as.err | cut -c1-300
tail -5 gpurun_out/r3as.err | cut -c1-300
echo "== dp tests"; timeout 900 python -m pytest tests/test_gpu_dp.py -q -m gpu 2>&1 | grep -E "passed|failed"
