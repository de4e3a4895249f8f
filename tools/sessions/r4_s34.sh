#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4_s34; mkdir -p $O
cd $R
timeout 600 python tools/probes/obs_f64.py > $O/obs_f64.jsonl 2> $O/err.log
echo done
