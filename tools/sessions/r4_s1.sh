#!/bin/bash
# Round-4 GPU session 1: tests, build A/B (r2 HEAD vs r3 HEAD vs worktree) on config 2, block sweep, context-effect counters.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4_s1; mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; echo "tests rc=$?" | tee -a $O/tests.log
timeout 600 python tools/ab_builds.py run --workload default --passes 4 > $O/ab_default.jsonl 2>&1
timeout 300 python tools/ab_builds.py run --workload stress --passes 2 > $O/ab_stress.jsonl 2>&1
timeout 300 python tools/ab_step.py default --out $O/geometry_default.jsonl > $O/geometry_default.log 2>&1
for st in fresh big_live big_freed big_freed_empty; do
  timeout 200 python tools/probes/context_pmc.py --state $st --time >> $O/context_times.jsonl 2>&1
done
cd /tmp; export TMPDIR=/tmp
for st in fresh big_freed_empty big_live; do
  i=0
  for set in "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_UTCL1_PERMISSION_MISS_sum" \
             "TCC_EA0_WRREQ TCC_EA0_WRREQ_STALL" \
             "TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_64B_sum GRBM_GUI_ACTIVE GRBM_UTCL2_BUSY" \
             "TCP_UTCL1_TRANSLATION_MISS_UNDER_MISS_sum TCP_UTCL1_STALL_INFLIGHT_MAX_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_WRITE_REQ_sum"; do
    i=$((i+1))
    timeout 300 rocprofv3 --pmc $set --output-format csv json -d $O/pmc_${st}_p$i -- python3 $R/tools/probes/context_pmc.py --state $st --steps 6 > $O/pmc_${st}_p$i.log 2>&1
    # the JSON carries per-instance values; keep only the fused step kernel's records, drop the rest (size)
    python3 $R/tools/sessions/slim_pmc.py $O/pmc_${st}_p$i >> $O/pmc_summary.jsonl 2>> $O/pmc_slim.err
  done
done
echo done
