#!/bin/bash
# Soak on the final sources: the GPU suite three times (fresh hypothesis / fuzz seeds each), then the fuzzers for ten minutes.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4_s40; mkdir -p $O
cd $R
for k in 1 2 3; do
  timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider > $O/tests_$k.log 2>&1; echo "run $k rc=$?" >> $O/summary.txt; tail -1 $O/tests_$k.log >> $O/summary.txt
done
( timeout 500 python tools/fuzz_variants.py 400 | tail -2 ) >> $O/summary.txt 2>&1
( timeout 200 python tools/fuzz_vec_env.py 120 | tail -1 ) >> $O/summary.txt 2>&1
( timeout 200 python tools/fuzz_dropin.py 90 | tail -1 ) >> $O/summary.txt 2>&1
echo done
