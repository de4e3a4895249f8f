#!/bin/bash
# Round-4 evidence session (kernel sources frozen; ran as r4_s7, s18, s28 and, finally, s31): tests, driver-like bench line, rocprofv3 evidence for every configuration the
# bench line quotes traffic for.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4_evidence; mkdir -p $O/profiles
cd $R
timeout 1500 python -m pytest tests -m gpu -q -x > $O/tests.log 2>&1; echo "tests rc=$?" | tee -a $O/tests.log
for spec in "stress linear 100" "stress table 2000 --no-export" "stress table 2000" "plugin table 2000" "default linear 4000" \
            "stress none 2000 --no-export --reward-per-env"; do
  timeout 900 bash tools/profile_bench.sh r4 $spec > $O/profile_$(echo $spec | tr ' ' '_' | tr -cd 'a-z0-9_').log 2>&1
done
cp profiles/r4_* $O/profiles/ 2>/dev/null
rm -rf $R/gpurun_out/prof_r4_*
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err
timeout 300 python bench.py --workload default --steps 2000 --warmup 2000 --no-cpu-baseline --no-single-env-latency > $O/bench_config2.json 2> $O/bench_config2.err
timeout 300 python bench.py --workload plugin --obs table --steps 1000 --warmup 1000 --no-cpu-baseline --no-single-env-latency > $O/bench_config4.json 2> $O/bench_config4.err

echo done
