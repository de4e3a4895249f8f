mkdir -p gpurun_out/r6i
for rep in 1 2 3; do
for t in "" "epw=2,sblock=128" "epw=2,sblock=256" "epw=8,sblock=512"; do
  python bench.py --workload default --steps 2000 --warmup 2000 --no-cpu-baseline --no-single-env-latency --no-extras ${t:+--tune $t} 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print(json.dumps({'tune':'$t' or 'default (4 envs x 64 threads per 256-thread workgroup)','avg_launch_us':round(r['avg_launch_ms']*1e3,2),'median_group_us':round((r.get('median_launch_ms') or 0)*1e3,2),'frac':round(r['frac'],3),'dist':{k:round(v,2) for k,v in (r.get('launch_distribution_us') or {}).items() if isinstance(v,float)}}))"
done; done > gpurun_out/r6i/config2_geometry.jsonl
cat gpurun_out/r6i/config2_geometry.jsonl
