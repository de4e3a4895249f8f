#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4_s39; mkdir -p $O
cd $R
D2D_BUILD_DIAG=1 python -m gym_d2d_amd.build --force > $O/build.log 2>&1
timeout 600 python tools/phase_times_small.py > $O/phase_times_small.json 2> $O/err.log
python -m gym_d2d_amd.build --force >> $O/build.log 2>&1
echo done
