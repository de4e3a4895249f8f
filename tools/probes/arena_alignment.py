#!/usr/bin/env python3
"""Follow-up of tools/probes/arena_map.py (every 2 MiB-aligned position inside a 48 GiB allocation is in the slow class) and
vmm_backing.py (an exact-size hipMalloc is fast 7 times of 8): is it the ALIGNMENT of the obs block's base?  One 2 GiB torch
arena; the 61 MB obs block bound at base + j x 4 KiB (j = 0 .. 63), + j x 64 KiB, + j x 1 MiB + small odd shifts; then eight
exact-size hipMallocs (through the library's own buffer) and eight torch.empty of the exact and of the rounded size."""
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


for k in range(3000):
    h.step(acts[k % 8].data_ptr())
arena = torch.empty(2 << 30, dtype=torch.uint8, device=env.device)
base = arena.data_ptr()
print(json.dumps({'arena_base_mod_2MiB': base % (2 << 20)}), flush=True)
for name, offs in (('j x 4 KiB', [j * 4096 for j in range(64)]), ('j x 64 KiB', [j * 65536 for j in range(64)]),
                   ('j x 256 B', [j * 256 for j in range(64)]), ('j x 1 MiB + (j % 8) x 16 KiB', [j * (1 << 20) + (j % 8) * 16384 for j in range(64)])):
    res = []
    for off in offs:
        h.bind_buffer(_native.BUF_OBS, base + off, nbytes)
        res.append(steady())
    print(json.dumps({'what': f'obs block at arena base + {name}', 'us_per_step': res}), flush=True)
del arena
torch.cuda.empty_cache()
held = []
for name, make in (('torch.empty of the exact 61 440 000 bytes (own segment, rounded up to 2 MiB by torch)', lambda: torch.empty(nbytes, dtype=torch.uint8, device=env.device)),
                   ('torch.empty of 64 MiB', lambda: torch.empty(64 << 20, dtype=torch.uint8, device=env.device)),
                   ('torch.empty of 61 440 000 + 4096 bytes', lambda: torch.empty(nbytes + 4096, dtype=torch.uint8, device=env.device))):
    res = []
    for k in range(8):
        t = make(); held.append(t)
        h.bind_buffer(_native.BUF_OBS, t.data_ptr(), nbytes)
        res.append(steady())
    print(json.dumps({'what': name, 'us_per_step': res, 'base_mod_2MiB': [t.data_ptr() % (2 << 20) for t in held[-8:]]}), flush=True)
env.close()
