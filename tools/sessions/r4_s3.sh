#!/bin/bash
# Round-4 GPU session 3: full GPU suite (ABI 3 + f64 obs), the placement probe.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4_s3; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -q > $O/tests.log 2>&1; echo "tests rc=$?" | tee -a $O/tests.log
hipcc --offload-arch=gfx950 -O3 tools/probes/placement.hip -o /tmp/placement > $O/placement_build.log 2>&1
timeout 600 /tmp/placement 3 > $O/placement.jsonl 2> $O/placement.err
echo done
