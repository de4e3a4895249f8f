#!/bin/bash
# Round-4 GPU session 4: tests, the seven allocation histories with dedicated blocks, a driver-like bench line, rocprofv3 evidence.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4_s4; mkdir -p $O/profiles
cd $R
timeout 1500 python -m pytest tests -m gpu -q -x > $O/tests.log 2>&1; echo "tests rc=$?" | tee -a $O/tests.log
timeout 600 python tools/probes/context_effect.py > $O/context_effect.jsonl 2> $O/context_effect.err
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err
for spec in "stress linear 100" "stress table 100 --no-export" "stress table 100" "plugin table 100" "default linear 200" \
            "stress none 100 --no-export --reward-per-env"; do
  timeout 900 bash tools/profile_bench.sh r4 $spec > $O/profile_$(echo $spec | tr ' ' '_' | tr -cd 'a-z0-9_').log 2>&1
done
cp profiles/r4_* $O/profiles/ 2>/dev/null
# raw traces are large: keep the summaries only
rm -rf $R/gpurun_out/prof_r4_*
echo done
