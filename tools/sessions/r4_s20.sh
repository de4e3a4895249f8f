#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4_s20; mkdir -p $O
cd $R
python -c "import __graft_entry__ as g; g.build(); g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/smoke.log
timeout 1500 python -m pytest tests -m gpu -q > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/tests.log
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err
timeout 300 python bench.py --workload default --steps 2000 --warmup 2000 --no-cpu-baseline --no-single-env-latency > $O/bench_config2.json 2> $O/bench_config2.err
timeout 300 python bench.py --workload plugin --obs table --steps 1000 --warmup 1000 --no-cpu-baseline --no-single-env-latency > $O/bench_config4.json 2> $O/bench_config4.err
echo done
