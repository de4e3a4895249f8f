#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4_s23; mkdir -p $O
cd $R
timeout 600 python tools/probes/arena_alignment.py > $O/arena_alignment.jsonl 2> $O/arena_alignment.err
echo done
