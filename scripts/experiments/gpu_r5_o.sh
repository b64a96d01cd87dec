#!/bin/bash
# round 5, call o: the multi-rank bench contract on one device: self re-exec of `python bench.py --gpus 2`, gloo + shared device, both exchanges
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
timeout 600 python bench.py --gpus 2 --steps 10 --warmup 3 --share-device --dist-backend gloo --no-cpu-baseline --no-harness 2> gpurun_out/r5o_a.err | tail -1 | cut -c1-700; tail -3 gpurun_out/r5o_a.err
timeout 600 python bench.py --gpus 2 --steps 10 --warmup 3 --share-device --dist-backend gloo --exchange flat --no-cpu-baseline --no-harness 2> gpurun_out/r5o_b.err | tail -1 | cut -c1-400; tail -3 gpurun_out/r5o_b.err
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29577 bench.py --gpus 2 --steps 10 --warmup 3 --share-device --dist-backend gloo --no-cpu-baseline --no-harness 2> gpurun_out/r5o_c.err | tail -1 | cut -c1-300; tail -2 gpurun_out/r5o_c.err
