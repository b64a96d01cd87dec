#!/bin/bash
# round 6, call A: the GPU suite on the re-gated refinit tests (all failures collected, no -x), the default bench line, and the
# K-runs-on-K-streams experiment
O=gpurun_out/r6a; mkdir -p $O gpurun_out/refinit
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err
timeout 600 python scripts/r6_dual_stream.py --batch 64 --runs 1,2,3,4 > $O/dual64.jsonl 2> $O/dual64.err
timeout 600 python scripts/r6_dual_stream.py --batch 128 --runs 1,2 > $O/dual128.jsonl 2> $O/dual128.err
timeout 600 python scripts/r6_dual_stream.py --batch 64 --runs 1,2 --method adapter > $O/dual64_adapter.jsonl 2> $O/dual64_adapter.err
tail -15 $O/pytest.log; cut -c1-400 $O/bench.json; cat $O/dual64.jsonl $O/dual128.jsonl $O/dual64_adapter.jsonl; tail -3 $O/dual64.err
