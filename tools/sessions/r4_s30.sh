#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4_s30; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -q -x > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/tests.log
for pass in 1 2 3; do
for mode in "none --reward-per-env --no-export" "table --no-export" "table"; do
for t in "" "walk=2"; do
  timeout 200 python bench.py --obs $mode --steps 2000 --warmup 1000 --no-cpu-baseline --no-single-env-latency --no-extras ${t:+--tune $t} 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.read()); print(json.dumps({'mode': '$mode', 'tune': '$t', 'us_per_launch': round(j['roofline']['avg_launch_ms'] * 1e3, 2)}))" >> $O/hot_lists_ab.jsonl
done
done
done
echo done
