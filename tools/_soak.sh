R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5_soak3; mkdir -p $O; cd $R
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider > $O/tests_2.log 2>&1; echo "second run rc=$? $(tail -1 $O/tests_2.log)" >> $O/summary.txt
( timeout 1000 python tools/fuzz_variants.py 900 | tail -2 ) >> $O/summary.txt 2>&1
( timeout 400 python tools/fuzz_vec_env.py 300 | tail -1 ) >> $O/summary.txt 2>&1
( timeout 400 python tools/fuzz_dropin.py 300 | tail -1 ) >> $O/summary.txt 2>&1
( timeout 300 python tools/fuzz_reset.py 200 | tail -1 ) >> $O/summary.txt 2>&1
cat $O/summary.txt
