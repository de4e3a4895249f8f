#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4_s8; mkdir -p $O
cd $R
timeout 600 python tools/probes/pool_ab.py > $O/pool_ab.jsonl 2> $O/pool_ab.err
timeout 900 python -m pytest tests/test_gpu_two_ranks.py tests/test_gpu_round4.py tests/test_gpu_round3.py -q -x > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/tests.log
echo done
