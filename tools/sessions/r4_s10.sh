#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4_s10; mkdir -p $O
cd $R
timeout 600 python tools/probes/obs_candidates.py > $O/obs_candidates.jsonl 2> $O/obs_candidates.err
timeout 600 python tools/probes/obs_candidates.py >> $O/obs_candidates.jsonl 2>> $O/obs_candidates.err
echo done
