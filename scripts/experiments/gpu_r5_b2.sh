#!/bin/bash
# round 5: the JSON line is the last line of stdout, also when RCCL has printed its banner (1-rank nccl group via --dp-route; two
# processes through torch.distributed.run)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
python bench.py --dp-route --steps 20 --no-cpu-baseline --no-harness 2>/dev/null | tail -1 | cut -c1-80
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29519 bench.py --gpus 2 --steps 10 --warmup 3 --dist-backend gloo --share-device 2>/dev/null | tail -1 | cut -c1-80
python bench.py --gpus 2 --steps 10 --warmup 3 --dist-backend gloo --share-device 2>/dev/null | tail -1 | cut -c1-80
python -m pytest tests/test_gpu_dp.py -q -k contract 2>&1 | tail -1
