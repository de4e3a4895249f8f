#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4_s21; mkdir -p $O
cd $R
timeout 900 python tools/probes/arena_map.py > $O/arena_map.jsonl 2> $O/arena_map.err
echo done
