#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4_s33; mkdir -p $O
cd $R
timeout 1200 python -m pytest tests/test_gpu_two_ranks.py -q -x > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/tests.log
echo done
