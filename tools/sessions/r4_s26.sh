#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4_s26; mkdir -p $O
cd $R
D2D_BUILD_DIAG=1 python -m gym_d2d_amd.build --force > $O/build.log 2>&1
timeout 600 python tools/elasticity.py --mode none --out $O/elasticity_none.json > $O/elasticity_none.log 2>&1
timeout 600 python tools/elasticity.py --mode table --out $O/elasticity_table.json > $O/elasticity_table.log 2>&1
python -m gym_d2d_amd.build --force >> $O/build.log 2>&1
echo done
