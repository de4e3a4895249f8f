#!/usr/bin/env python3
"""Speed class of BASELINE config 2 against the SIZE and the VIRTUAL ALIGNMENT of the allocation its 61 MB obs block is bound
to (each allocation its own torch segment = its own hipMalloc, all held): the page-table fragment the driver can use grows with
both, and every piece of a large allocation has been in the slow class (tools/probes/arena_map.py)."""
import json
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import torch

from gym_d2d_amd import _native
from gym_d2d_amd.envs import VecD2DEnv

env = VecD2DEnv({'num_rbs': 25, 'num_cues': 25, 'num_due_pairs': 25}, num_envs=1024, cue_actions='traffic', placement_trials=0)
env.reset(seed=1)
h = env.simulator.handle
acts = torch.randint(0, 25 * 21, (8, 1024, 25), device=env.device, dtype=torch.int32)
nbytes = 1024 * 50 * 300 * 4


def steady(steps=500):
    for k in range(60):
        h.step(acts[k % 8].data_ptr())
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(steps):
        h.step(acts[k % 8].data_ptr())
    torch.cuda.synchronize()
    return round((time.perf_counter() - t0) / steps * 1e6, 2)


def align_mib(p):                 # largest power of two (MiB, capped at 1024) dividing the address
    a = 1
    while a < 1024 and p % ((a * 2) << 20) == 0:
        a *= 2
    return a


for k in range(3000):
    h.step(acts[k % 8].data_ptr())
held = []
for rnd in range(2):
    for mib in (60, 62, 64, 66, 96, 128, 59, 61):
        res, al = [], []
        for k in range(8):
            t = torch.empty(mib << 20, dtype=torch.uint8, device=env.device); held.append(t)
            h.bind_buffer(_native.BUF_OBS, t.data_ptr(), nbytes)
            res.append(steady()); al.append(align_mib(t.data_ptr()))
        print(json.dumps({'round': rnd, 'allocation_MiB': mib, 'us_per_step': res, 'va_alignment_MiB': al}), flush=True)
env.close()
