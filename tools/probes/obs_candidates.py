#!/usr/bin/env python3
"""Is WHERE the 61 MB obs block sits what puts BASELINE config 2 into its 13.1 / 13.9 / 15.0 us classes?  One env, one set of
every other buffer; 16 candidate obs blocks held at once (distinct physical ranges), bound one after the other, 1500 steps
each; then the same for the other per-step buffers together (table, planes, reward, rb, pwr) with the obs block fixed."""
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
acts = torch.randint(0, 25 * 21, (8, 1024, 25), device=env.device, dtype=torch.int32)


def steady(steps=1500):
    for k in range(500):
        h.step(acts[k % 8].data_ptr())
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(steps):
        h.step(acts[k % 8].data_ptr())
    torch.cuda.synchronize()
    return round((time.perf_counter() - t0) / steps * 1e6, 2)


print(json.dumps({'what': 'as created', 'us_per_step': steady()}), flush=True)
n = 1024 * 50 * 300
cands = [torch.empty(n, dtype=torch.float32, device=env.device) for _ in range(16)]
for rnd in range(2):
    res = []
    for c in cands:
        h.bind_buffer(_native.BUF_OBS, c.data_ptr(), n * 4)
        res.append(steady())
    print(json.dumps({'what': 'obs block = candidate k of 16 (held at once)', 'round': rnd, 'us_per_step': res,
                      'GiB_offsets': [round((c.data_ptr() - cands[0].data_ptr()) / 2**30, 3) for c in cands]}), flush=True)
best = min(range(16), key=lambda k: res[k])
h.bind_buffer(_native.BUF_OBS, cands[best].data_ptr(), n * 4)
small = {'table': (_native.BUF_OBS_TABLE, 1024 * 50 * 6), 'sinr': (_native.BUF_SINR_DB, 1024 * 50), 'snr': (_native.BUF_SNR_DB, 1024 * 50),
         'rate': (_native.BUF_RATE_BPS, 1024 * 50), 'cap': (_native.BUF_CAPACITY, 1024 * 50), 'reward': (_native.BUF_REWARD, 1024 * 50),
         'rb': (_native.BUF_RB, 1024 * 50), 'pwr': (_native.BUF_PWR, 1024 * 50)}
res = []
keep = []
for k in range(8):
    pad = torch.empty((k + 1) * 3_000_000, dtype=torch.uint8, device=env.device)        # shifts where the next ones land
    ts = {name: torch.empty(words, dtype=torch.float32, device=env.device) for name, (_, words) in small.items()}
    keep.append((pad, ts))
    for name, (which, words) in small.items():
        h.bind_buffer(which, ts[name].data_ptr(), words * 4)
    res.append(steady())
print(json.dumps({'what': 'best obs candidate fixed; the other per-step buffers re-allocated 8 times', 'us_per_step': res}), flush=True)
env.close()
