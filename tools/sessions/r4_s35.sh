#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4_s35; mkdir -p $O
cd $R
for st in fresh after_headline fresh after_headline; do
  timeout 300 python tools/probes/trial_budget.py $st >> $O/trial_budget.jsonl 2>> $O/err.log
done
echo done
