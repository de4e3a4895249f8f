#!/bin/bash
# Regenerates the round's evidence on ONE GPU box (run through gpurun from the repo root):
#   gpurun --timeout 3000 -- 'bash tools/evidence.sh r6'
# What it leaves under gpurun_out/<tag>_evidence/ (the only directory gpurun brings back): profiles/<tag>_kernel_stats_*.csv,
# profiles/<tag>_pmc_*.json (rocprofv3 kernel trace and PMC passes, collected in SEPARATE runs, stamped with the digest of the
# kernel sources), profiles/<tag>_bench_lines.jsonl (the driver-like default line and the stand-alone configurations), the GPU
# suite's log.  Then, here:  cp gpurun_out/<tag>_evidence/profiles/* profiles/
tag=${1:-r6}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${tag}_evidence; mkdir -p $O; cd $R
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider > $O/tests.log 2>&1; echo "gpu tests rc=$? $(tail -1 $O/tests.log)" > $O/summary.txt
# rocprofv3 first: kernel trace + PMC passes per workload (tools/profile_bench.sh); bench.py quotes `roofline.traffic` from the
# summaries of THESE sources (digest match), so the lines below all carry it
bash tools/profile_bench.sh $tag stress linear 100 > $O/prof_stress_linear.log 2>&1
bash tools/profile_bench.sh $tag stress linear 30 --obs-dtype float64 > $O/prof_stress_linear_f64.log 2>&1
bash tools/profile_bench.sh $tag stress table 1000 --no-export > $O/prof_stress_table.log 2>&1
bash tools/profile_bench.sh $tag stress table 1000 > $O/prof_stress_table_export.log 2>&1
bash tools/profile_bench.sh $tag stress none 1000 --no-export --reward-per-env > $O/prof_stress_none.log 2>&1
bash tools/profile_bench.sh $tag default linear 4000 > $O/prof_default_linear.log 2>&1
bash tools/profile_bench.sh $tag plugin table 1000 > $O/prof_plugin_table.log 2>&1
# the power-law kernels (COST-Hata: an exponent other than 2), obs-less and with the compact table
bash tools/profile_bench.sh $tag hata none 1000 --no-export --reward-per-env > $O/prof_hata_none.log 2>&1
bash tools/profile_bench.sh $tag hata table 1000 --no-export > $O/prof_hata_table.log 2>&1
# the driver-like line (headline + core_mode + other_workloads + flat scalars), then the stand-alone configurations
: > profiles/${tag}_bench_lines.jsonl
for args in "" "--workload default" "--workload plugin" "--obs table --no-export" "--obs table" \
            "--obs none --no-export --reward-per-env" "--obs-dtype float64 --steps 30" \
            "--workload hata --obs none --no-export --reward-per-env" "--workload hata --obs table --no-export"; do
  extra="--no-cpu-baseline --no-single-env-latency --no-extras"; [ -z "$args" ] && extra=""
  timeout 1200 python bench.py $args $extra 2>> $O/bench.err | tail -1 >> profiles/${tag}_bench_lines.jsonl
done
python3 - "$tag" <<'PY'
import json, sys
for l in open(f'profiles/{sys.argv[1]}_bench_lines.jsonl'):
    if not l.strip():
        continue
    d = json.loads(l)
    r = d['roofline']
    print(d['config']['workload'][:40], d['config'].get('obs_mode'), d.get('obs_dtype'), 'ms/step %.4f' % d['ms_per_step'], r['kernel'],
          'frac %.3f' % r['frac'], 'median frac', r.get('frac_at_median_launch'), 'traffic', r.get('traffic'))
PY
mkdir -p $O/profiles; cp profiles/${tag}_kernel_stats_* profiles/${tag}_pmc_* profiles/${tag}_bench_lines.jsonl $O/profiles/
cat $O/summary.txt; ls $O/profiles
