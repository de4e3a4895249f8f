#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4_s38; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_round4.py -q -x -k "numpy_path or obs_less or placement" > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/tests.log
echo done
