#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4_s6; mkdir -p $O
cd $R
timeout 300 python tools/probes/clock_state.py > $O/clock_state.jsonl 2> $O/clock_state.err
timeout 900 python tools/probes/obs_policy_geometry.py > $O/obs_policy_geometry.jsonl 2> $O/obs_policy_geometry.err
timeout 300 python tools/probes/clock_state.py >> $O/clock_state.jsonl 2>> $O/clock_state.err
echo done
