#!/usr/bin/env python3
"""Does the placement lottery of tools/probes/obs_candidates.py also decide the compact-obs step at 4096 x 512 (config 4: 29.0
in the bench line against 26.8 us stand-alone on one box)?  One env; 8 candidate SETS of the step's output buffers (table 50 MB,
four result planes, reward, rb, pwr: 106 MB per set), all held at once, bound one after the other."""
import json
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import torch

from gym_d2d_amd import _native
from gym_d2d_amd.envs import VecD2DEnv
from gym_d2d_amd.envs.obs_fn import OwnLinkObsFunction

B, N = 4096, 512
env = VecD2DEnv({'num_rbs': 256, 'num_cues': 256, 'num_due_pairs': 256, 'obs_fn': OwnLinkObsFunction}, num_envs=B)
env.reset(seed=1)
h = env.simulator.handle
acts = torch.randint(0, 256 * 21, (64, B, N), device=env.device, dtype=torch.int32)


def steady(steps=1500):
    for k in range(700):
        h.step(acts[k % 64].data_ptr())
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(steps):
        h.step(acts[k % 64].data_ptr())
    torch.cuda.synchronize()
    return round((time.perf_counter() - t0) / steps * 1e6, 2)


bufs = {'table': (_native.BUF_OBS_TABLE, B * N * 6), 'sinr': (_native.BUF_SINR_DB, B * N), 'snr': (_native.BUF_SNR_DB, B * N),
        'rate': (_native.BUF_RATE_BPS, B * N), 'cap': (_native.BUF_CAPACITY, B * N), 'reward': (_native.BUF_REWARD, B * N),
        'rb': (_native.BUF_RB, B * N), 'pwr': (_native.BUF_PWR, B * N)}
print(json.dumps({'what': 'as created', 'us_per_step': steady()}), flush=True)
sets = [{name: torch.empty(words, dtype=torch.float32, device=env.device) for name, (_, words) in bufs.items()} for _ in range(8)]
for rnd in range(2):
    res = []
    for ts in sets:
        for name, (which, words) in bufs.items():
            h.bind_buffer(which, ts[name].data_ptr(), words * 4)
        res.append(steady())
    print(json.dumps({'what': 'all output buffers = candidate set k of 8', 'round': rnd, 'us_per_step': res}), flush=True)
# one buffer at a time, the rest from set 0
for name, (which, words) in bufs.items():
    for n2, (w2, words2) in bufs.items():
        h.bind_buffer(w2, sets[0][n2].data_ptr(), words2 * 4)
    res = []
    for ts in sets:
        h.bind_buffer(which, ts[name].data_ptr(), words * 4)
        res.append(steady(800))
    print(json.dumps({'what': f'only `{name}` varies over the 8 sets', 'us_per_step': res}), flush=True)
env.close()
