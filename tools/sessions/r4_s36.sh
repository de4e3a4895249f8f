#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4_s36; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_round4.py tests/test_gpu_round2.py -q -x > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/tests.log
timeout 900 python bench.py --no-cpu-baseline > $O/bench_default.json 2> $O/bench_default.err
echo done
