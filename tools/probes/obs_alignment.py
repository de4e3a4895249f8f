#!/usr/bin/env python3
"""Does the fused small-N step's time depend on WHERE its obs buffer sits?  One 1 GiB arena, the 61 MB obs block bound at
different offsets inside it (and the same offsets in a second arena), time per step for each."""
import json
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import torch

from gym_d2d_amd import _native
from gym_d2d_amd.envs import VecD2DEnv

env = VecD2DEnv({'num_rbs': 25, 'num_cues': 25, 'num_due_pairs': 25}, num_envs=1024, cue_actions='traffic')
env.reset(seed=1)
h = env.simulator.handle
acts = torch.randint(0, 25 * 21, (220, 1024, 25), device=env.device, dtype=torch.int32)
nbytes = 1024 * 50 * 300 * 4


def measure():
    res = []
    for rnd in range(2):
        for k in range(20):
            h.step(acts[k].data_ptr())
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(20, 220):
            h.step(acts[k].data_ptr())
        torch.cuda.synchronize()
        res.append((time.perf_counter() - t0) / 200 * 1e6)
    return round(min(res), 2)


print(json.dumps({'where': 'the env\'s own torch tensor', 'ptr_mod_2MiB': env._t['obs'].data_ptr() % (2 << 20), 'us_per_step': measure()}), flush=True)
for arena_id in range(3):
    arena = torch.empty(1 << 30, dtype=torch.uint8, device=env.device)
    base = arena.data_ptr()
    for off in (0, 512, 4096, 65536, 1 << 20, 2 << 20, (2 << 20) + 4096, 64 << 20, (64 << 20) + 245760, 300 << 20, (300 << 20) + 512):
        h.bind_buffer(_native.BUF_OBS, base + off, nbytes)
        print(json.dumps({'arena': arena_id, 'arena_base_mod_2MiB': base % (2 << 20), 'offset': off, 'us_per_step': measure()}), flush=True)
    hold = torch.empty((arena_id + 1) * 123456789, dtype=torch.uint8, device=env.device)   # move the next arena somewhere else
