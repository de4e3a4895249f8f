#!/usr/bin/env python3
"""Speed class of BASELINE config 2's fused step as a function of WHERE inside one 48 GiB allocation its 61 MB obs block is bound:
offsets every 64 MiB across the arena, then every 2 MiB across 512 MiB around a fast and a slow spot.  Is there an address
structure (a period, an alignment) behind the placement classes of tools/probes/obs_candidates.py?"""
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


def steady(steps=600):
    for k in range(60):
        h.step(acts[k % 8].data_ptr())
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(steps):
        h.step(acts[k % 8].data_ptr())
    torch.cuda.synchronize()
    return round((time.perf_counter() - t0) / steps * 1e6, 2)


for k in range(3000):                       # past the clock ramp
    h.step(acts[k % 8].data_ptr())
arena = torch.empty(48 << 30, dtype=torch.uint8, device=env.device)
base = arena.data_ptr()
print(json.dumps({'arena_base_mod_1GiB_MiB': (base % (1 << 30)) >> 20, 'arena_GiB': 48}), flush=True)
coarse = []
for k in range(0, 760):
    off = k * (64 << 20)
    h.bind_buffer(_native.BUF_OBS, base + off, nbytes)
    coarse.append(steady())
print(json.dumps({'what': 'obs block at arena offset k x 64 MiB, k = 0 .. 759', 'us_per_step': coarse}), flush=True)
fast = min(range(len(coarse)), key=coarse.__getitem__)
slow = max(range(len(coarse)), key=coarse.__getitem__)
for name, centre in (('around the fastest 64 MiB spot', fast), ('around the slowest 64 MiB spot', slow)):
    start = max(0, centre * (64 << 20) - (256 << 20))
    fine = []
    for k in range(256):
        h.bind_buffer(_native.BUF_OBS, base + start + k * (2 << 20), nbytes)
        fine.append(steady(400))
    print(json.dumps({'what': f'{name} (k = {centre}): offsets start + k x 2 MiB, k = 0 .. 255', 'start_MiB': start >> 20, 'us_per_step': fine}), flush=True)
env.close()
