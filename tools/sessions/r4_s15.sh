#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4_s15; mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_round4.py -q -x -k "packed" > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/tests.log
D2D_X=1 timeout 600 python tools/probes/obs_candidates.py > $O/obs_candidates.jsonl 2> $O/obs_candidates.err
echo done
