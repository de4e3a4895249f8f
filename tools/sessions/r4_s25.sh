#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4_s25; mkdir -p $O
cd $R
( timeout 400 python tools/fuzz_variants.py 300 | tail -3 ) > $O/fuzz.txt 2>&1
( timeout 200 python tools/fuzz_vec_env.py 120 | tail -2 ) >> $O/fuzz.txt 2>&1
( timeout 200 python tools/fuzz_dropin.py 90 | tail -2 ) >> $O/fuzz.txt 2>&1
( timeout 100 python tools/fuzz_reset.py 40 | tail -2 ) >> $O/fuzz.txt 2>&1
( time python bench.py > $O/bench_default.json 2> $O/bench_default.err ) 2> $O/bench_time.txt
echo done
