#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4_s32; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -q > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/tests.log
echo done
