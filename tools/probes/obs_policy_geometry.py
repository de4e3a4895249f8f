#!/usr/bin/env python3
"""The obs kernel's store policy x launch geometry, interleaved rounds in one process on the real 4096 x 512 workload
(follow-up of tools/probes/store_policy.py: a staged 1024-thread fill with `sc1 nt` stores reached 7.45 TB/s)."""
import json
import statistics
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import torch

from gym_d2d_amd import _native
from gym_d2d_amd.envs import VecD2DEnv

NAMES = {1: 'nt', 5: 'sc1 nt', 4: 'sc0 sc1 nt', 0: 'plain'}
GEOMS = [(0, 0, 0), (768, 2, 3), (960, 2, 2), (896, 2, 2), (832, 2, 2), (1024, 3, 2), (704, 3, 2), (640, 3, 2), (1024, 2, 2)]      # (block, rows or passes, variant: 2 = flat slabs)
env = VecD2DEnv({'num_rbs': 256, 'num_cues': 256, 'num_due_pairs': 256}, num_envs=4096)
env.reset(seed=1)
h = env.simulator.handle
act = torch.randint(0, 256 * 21, (8, 4096, 512), device=env.device, dtype=torch.int32)
ref = None
times = {}
for rnd in range(5):
    for block, rows, variant in GEOMS:
        for p in NAMES:
            if p in (0, 4) and (block, rows) != (0, 0):
                continue
            h.set_tuning(_native.TUNE_OBS_VARIANT, variant)
            h.set_tuning(_native.TUNE_OBS_NONTEMPORAL, p)
            h.set_tuning(_native.TUNE_OBS_BLOCK, block)
            h.set_tuning(_native.TUNE_OBS_ROWS_PER_WG, rows)
            for k in range(2):
                h.step(act[k % 8].data_ptr())
            h.profile_reset(); h.profile_enable(True)
            for k in range(10):
                h.step(act[k % 8].data_ptr())
            ms, n = h.profile_read(1)
            h.profile_enable(False)
            times.setdefault((block, rows, variant, p), []).append(ms / n)
            if rnd == 0:                                  # every variant writes the same bits
                h.step(act[0].data_ptr())
                torch.cuda.synchronize()
                chk = env._t['obs'][::97, ::31].clone()
                if ref is None:
                    ref = chk
                assert torch.equal(ref, chk), (block, rows, variant, p)
bytes_per_launch = 4096 * 512 * (24 * 512 + 24)
for (block, rows, variant, p), t in times.items():
    print(json.dumps({'kernel': 'obs_expand_flat_kernel (flat slabs: passes of `block` float4)' if variant in (0, 2) else 'obs_expand_kernel (row-aligned, rounds 1-3)', 'block': block or 1024,
                      'rows_or_passes_per_wg': rows or 2, 'store_policy': NAMES[p],
                      'median_ms': round(statistics.median(t), 4), 'min_ms': round(min(t), 4),
                      'median_GBps': round(bytes_per_launch / statistics.median(t) / 1e6, 1)}), flush=True)
env.close()
