#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4_s16; mkdir -p $O
cd $R
timeout 900 python tools/probes/obs_policy_geometry.py > $O/obs_policy_geometry.jsonl 2> $O/obs_policy_geometry.err
echo done
