R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5_s7; mkdir -p $O; cd $R
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider > $O/tests.log 2>&1; echo "tests rc=$?" > $O/summary.txt; tail -8 $O/tests.log >> $O/summary.txt
for m in none table; do
  timeout 300 python tools/ab_builds.py run --workload stress --mode $m --no-export --passes 3 >> $O/ab.jsonl 2>&1
done
timeout 300 python tools/ab_builds.py run --workload stress --mode table --passes 3 >> $O/ab.jsonl 2>&1
timeout 300 python tools/ab_builds.py run --workload default --passes 2 >> $O/ab.jsonl 2>&1
cat $O/summary.txt; cat $O/ab.jsonl
