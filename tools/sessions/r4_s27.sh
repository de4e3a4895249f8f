#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4_s27; mkdir -p $O
cd $R
timeout 900 python tools/probes/walk_divergence_bound.py > $O/walk_divergence_bound.jsonl 2> $O/err.log
echo done
