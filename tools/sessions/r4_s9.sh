#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4_s9; mkdir -p $O
cd $R
timeout 600 python tools/probes/history_ab.py A > $O/history_ab.jsonl 2> $O/history_ab.err
timeout 600 python tools/probes/history_ab.py B >> $O/history_ab.jsonl 2>> $O/history_ab.err
echo done
