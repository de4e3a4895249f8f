R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5_s11; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_gpu_step_variants.py -m gpu -q -p no:cacheprovider > $O/tests.log 2>&1; echo "tests rc=$?" > $O/summary.txt; tail -3 $O/tests.log >> $O/summary.txt
timeout 600 python tools/ab_builds.py run --workload stress --mode none --no-export --passes 3 >> $O/ab.jsonl 2>&1
timeout 600 python tools/ab_builds.py run --workload stress --mode table --no-export --passes 2 >> $O/ab.jsonl 2>&1
cat $O/summary.txt $O/ab.jsonl
