#!/bin/bash
# round 5: the pipelined DP step (exchange + SGD on a second stream under the next step's stem): tests, the three routes on a 1-rank
# RCCL group, two processes sharing the device
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
python -m pytest tests/test_gpu_dp.py tests/test_gpu_allreduce.py -x -q 2>&1 | grep -E "passed|failed|Error|error" | tail -5
for b in 128 64; do
python bench.py --dp-route --batch $b --no-cpu-baseline --no-harness 2>/dev/null | grep '^{"metric' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(json.dumps({k: v for k, v in d['dp_route'].items() if k != 'how'})); print(d['value'], d['ms_per_step'])"
done
for ex in single pipelined staged; do
  python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 20 --warmup 5 --dist-backend gloo --share-device --dp-exchange $ex 2>/dev/null | grep '^{"metric' | cut -c1-200
done
