#!/bin/bash
# Round-4 GPU session 2: new tests, the slow allocation state under counters, the write-gap probes, planes_only, a bench line.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4_s2; mkdir -p $O
cd $R
timeout 1200 python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; echo "tests rc=$?" | tee -a $O/tests.log
for st in fresh big_live big_freed big_freed_empty; do
  timeout 200 python tools/probes/context_pmc.py --state $st --time >> $O/context_times.jsonl 2>&1
done
for own in obs all; do
  timeout 200 python tools/probes/context_pmc.py --state big_freed --time --own $own >> $O/context_times.jsonl 2>&1
done
for rot in 0 1 7 53; do
  timeout 200 python tools/probes/context_pmc.py --state big_freed --time --rotate $rot >> $O/context_times.jsonl 2>&1
done
timeout 900 python tools/probes/write_gap.py all > $O/write_gap.jsonl 2> $O/write_gap.err
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err
cd /tmp; export TMPDIR=/tmp
for st in big_freed fresh; do
  i=0
  for set in "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_UTCL1_PERMISSION_MISS_sum" \
             "TCC_EA0_WRREQ TCC_EA0_WRREQ_STALL" \
             "TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_64B_sum GRBM_GUI_ACTIVE GRBM_UTCL2_BUSY" \
             "TCP_UTCL1_TRANSLATION_MISS_UNDER_MISS_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_NC_WRITE_REQ_sum" \
             "TCC_HIT_sum TCC_MISS_sum TCC_WRITEBACK_sum TCC_EA0_WRREQ_LEVEL_sum"; do
    i=$((i+1))
    timeout 300 rocprofv3 --pmc $set --output-format csv json -d $O/pmc_${st}_p$i -- python3 $R/tools/probes/context_pmc.py --state $st --steps 6 > $O/pmc_${st}_p$i.log 2>&1
    python3 $R/tools/sessions/slim_pmc.py $O/pmc_${st}_p$i >> $O/pmc_summary.jsonl 2>> $O/pmc_slim.err
  done
done
echo done
