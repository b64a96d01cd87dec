#!/bin/bash
# round 5: the single-exchange DP route (fused call + one all-reduce of the flat buffer) against the staged one: tests, and both on a
# 1-rank RCCL group beside the fused step (bench.py --dp-route); two processes sharing the device through torch.distributed.run
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
python -m pytest tests/test_gpu_dp.py tests/test_gpu_allreduce.py -x -q 2>&1 | grep -E "passed|failed|Error|error" | tail -5
python bench.py --dp-route --no-cpu-baseline --no-harness 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(json.dumps(d['dp_route'], indent=1)); print(d['value'], d['ms_per_step'])"
python bench.py --dp-route --batch 64 --no-cpu-baseline --no-harness 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(json.dumps({k: v for k, v in d['dp_route'].items() if k != 'how'})); print(d['value'], d['ms_per_step'])"
for ex in single staged; do
  python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 20 --warmup 5 --dist-backend gloo --share-device --dp-exchange $ex 2>/dev/null | tail -1 | cut -c1-200
done
