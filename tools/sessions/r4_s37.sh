#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4_s37; mkdir -p $O
cd $R
( time python bench.py --gpus 1 --steps 20 --warmup 3 > $O/bench_20.json 2> $O/bench_20.err ) 2> $O/time_20.txt
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/smoke.log
echo done
