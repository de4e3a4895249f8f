#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4_s12; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -q -x > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/tests.log
timeout 600 python tools/probes/history_ab.py A > $O/history_ab_A.jsonl 2> $O/history_ab.err
timeout 600 python tools/probes/history_ab.py B > $O/history_ab_B.jsonl 2>> $O/history_ab.err
timeout 600 python tools/probes/context_effect.py > $O/context_effect.jsonl 2>> $O/history_ab.err
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err
echo done
