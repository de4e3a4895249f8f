#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4_s13; mkdir -p $O
cd $R
timeout 900 python bench.py --no-cpu-baseline > $O/bench_default.json 2> $O/bench_default.err
timeout 600 python tools/probes/history_ab.py A > $O/history_ab_A.jsonl 2> $O/history_ab.err
timeout 600 python tools/probes/history_ab.py B > $O/history_ab_B.jsonl 2>> $O/history_ab.err
timeout 900 python bench.py --no-cpu-baseline > $O/bench_default2.json 2> $O/bench_default2.err
echo done
