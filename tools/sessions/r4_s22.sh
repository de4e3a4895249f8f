#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4_s22; mkdir -p $O
cd $R
hipcc --offload-arch=gfx950 -O2 -shared -fPIC tools/probes/vmm_alloc.hip -o /tmp/libvmm_alloc.so > $O/build.log 2>&1
timeout 900 python tools/probes/vmm_backing.py > $O/vmm_backing.jsonl 2> $O/vmm_backing.err
echo done
