#!/bin/bash
# Is the member-list search cheaper than the mask walk now that the obs-less step sits at its VALU floor?
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4_s19; mkdir -p $O
cd $R
C="--obs none --reward-per-env --no-export --steps 2000 --warmup 1000 --no-cpu-baseline --no-single-env-latency --no-extras"
for pass in 1 2; do
for t in "" "walk=2" "prefetch=0" "lpt=2" "lpt=2,walk=2"; do
  timeout 200 python bench.py $C ${t:+--tune $t} 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.read()); print(json.dumps({'tune': '$t', 'obs': 'none', 'us_per_launch': round(j['roofline']['avg_launch_ms'] * 1e3, 2), 'ms_per_step': j['ms_per_step']}))" >> $O/walk_ab.jsonl
done
done
C="--obs table --no-export --steps 2000 --warmup 1000 --no-cpu-baseline --no-single-env-latency --no-extras"
for t in "" "walk=2" "prefetch=0"; do
  timeout 200 python bench.py $C ${t:+--tune $t} 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.read()); print(json.dumps({'tune': '$t', 'obs': 'table', 'us_per_launch': round(j['roofline']['avg_launch_ms'] * 1e3, 2), 'ms_per_step': j['ms_per_step']}))" >> $O/walk_ab.jsonl
done
echo done
