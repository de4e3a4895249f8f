#!/bin/bash
# rocprofv3 evidence for one bench workload, on the GPU box (counters and traces in SEPARATE runs):
#   tools/profile_bench.sh <tag> <workload> <obs> [steps] [extra bench args]     e.g.  tools/profile_bench.sh r3 stress table 100 --no-export
# Writes raw output under gpurun_out/prof_<tag>_<workload>_<obs>/ and the condensed summaries into profiles/.
tag=$1; wl=$2; obs=$3; steps=${4:-100}; extra="${@:5}"
R=$GRAFT_REPO_ROOT
sfx=$(echo "$extra" | tr -cd 'a-z')
out=$R/gpurun_out/prof_${tag}_${wl}_${obs}${sfx:+_$sfx}
mkdir -p $out
cd /tmp; export TMPDIR=/tmp
wu=10; [ "$steps" -gt 500 ] && wu=$((steps / 2))      # short kernels: long enough to be past the clock ramp behind an idle stretch
args="--workload $wl --obs $obs --no-cpu-baseline --no-single-env-latency --no-extras $extra"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt -- python3 $R/bench.py $args --steps $steps --warmup $wu > $out/kt.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -- python3 $R/bench.py $args --steps 5 --warmup 1 > $out/pmc_write.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -- python3 $R/bench.py $args --steps 5 --warmup 1 > $out/pmc_fetch.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d $out/pmc_sq1 -- python3 $R/bench.py $args --steps 5 --warmup 1 > $out/pmc_sq1.log 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS --output-format csv -d $out/pmc_sq2 -- python3 $R/bench.py $args --steps 5 --warmup 1 > $out/pmc_sq2.log 2>&1
cd $R
PROFILE_WARMUP=$wu python3 tools/summarize_profiles.py $tag $wl $obs $out $extra
