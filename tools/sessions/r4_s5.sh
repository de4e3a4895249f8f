#!/bin/bash
# Round-4 GPU session 5: same-box A/B of who allocates the env's buffers (torch.empty / dedicated blocks / obs only) across
# allocation histories; store cache-policy probe.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4_s5; mkdir -p $O
cd $R
for pass in 1 2; do
for owner in torch block obs; do
  for st in fresh big_live big_freed big_freed_empty; do
    D2D_VEC_ENV_BUFFERS=$owner timeout 200 python tools/probes/context_pmc.py --state $st --time 2>/dev/null | sed "s/^{/{\"buffers\": \"$owner\", /" >> $O/context_owner_ab.jsonl
  done
done
done
timeout 900 python tools/probes/store_policy.py > $O/store_policy.jsonl 2> $O/store_policy.err
echo done
