#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4_s11; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -q -x > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/tests.log
timeout 900 python tools/probes/out_candidates_stress.py > $O/out_candidates_stress.jsonl 2> $O/out_candidates_stress.err
timeout 600 python tools/probes/history_ab.py A > $O/history_ab_with_trials.jsonl 2> $O/history_ab.err
echo done
