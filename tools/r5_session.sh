R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5_s1; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_round3.py tests/test_gpu_round4.py tests/test_gpu_round2.py tests/test_gpu_parity.py -m gpu -x -q -p no:cacheprovider > $O/tests.log 2>&1; echo "tests rc=$?" > $O/summary.txt; tail -3 $O/tests.log >> $O/summary.txt
for m in none table; do for w in 2 -1; do
  timeout 300 python tools/ab_builds.py run --workload stress --mode $m --walk $w --no-export --passes 2 >> $O/ab.jsonl 2>&1
done; done
timeout 300 python tools/ab_builds.py run --workload stress --mode table --walk 2 --passes 2 >> $O/ab.jsonl 2>&1
cat $O/summary.txt; cat $O/ab.jsonl
