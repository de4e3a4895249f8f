#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4_s24; mkdir -p $O
cd $R
timeout 600 python tools/probes/alloc_size_alignment.py > $O/alloc_size_alignment.jsonl 2> $O/err.log
echo done
