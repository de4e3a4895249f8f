#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4_s14; mkdir -p $O
cd $R
for v in nothing single_env_first headline_short headline_full headline_core; do
  timeout 400 python tools/probes/bench_bisect.py $v >> $O/bisect.jsonl 2>> $O/bisect.err
done
echo done
