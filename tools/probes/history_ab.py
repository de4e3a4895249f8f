#!/usr/bin/env python3
"""What about the process's history slows BASELINE config 2 from 13.3 to 15.0 us per step (steady state, long runs)?
One variable at a time, in one process, after a 26 GB env was stepped and closed."""
import json
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import torch

from gym_d2d_amd import _native
from gym_d2d_amd.envs import VecD2DEnv


def small(tag, own_obs=False):
    env = VecD2DEnv({'num_rbs': 25, 'num_cues': 25, 'num_due_pairs': 25}, num_envs=1024, cue_actions='traffic')
    if own_obs:
        env.simulator.handle.bind_buffer(_native.BUF_OBS, 0, 0)         # the library's own hipMalloc instead of the torch tensor
    env.reset(seed=1)
    acts = torch.randint(0, 25 * 21, (8, 1024, 25), device=env.device, dtype=torch.int32)
    for k in range(2000):
        env.step(acts[k % 8])
    res = []
    for c in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(1000):
            env.step(acts[k % 8])
        torch.cuda.synchronize()
        res.append(round((time.perf_counter() - t0) / 1000 * 1e6, 2))
    free, total = torch.cuda.mem_get_info()
    print(json.dumps({'state': tag, 'own_obs': own_obs, 'us_per_step': res, 'torch_reserved_GB': round(torch.cuda.memory_reserved() / 1e9, 2),
                      'device_used_GB': round((total - free) / 1e9, 2)}), flush=True)
    env.close()


def big_env():
    big = VecD2DEnv({'num_rbs': 256, 'num_cues': 256, 'num_due_pairs': 256}, num_envs=4096)
    big.reset(seed=1)
    a = torch.randint(0, 256 * 21, (4096, 512), device=big.device, dtype=torch.int32)
    for k in range(5):
        big.step(a)
    torch.cuda.synchronize()
    big.close()


order = sys.argv[1] if len(sys.argv) > 1 else 'A'
small('fresh process')
if order == 'A':
    big_env()
    small('after a closed 26 GB env (blocks in torch cache)')
    small('same, obs owned by the library', own_obs=True)
    torch.cuda.empty_cache()
    small('after torch.cuda.empty_cache()')
    x = torch.empty(int(26e9), dtype=torch.uint8, device='cuda'); del x
    small('a 26 GB tensor allocated, never touched, freed to the cache')
    torch.cuda.empty_cache()
    small('empty_cache again')
else:
    x = torch.empty(int(26e9), dtype=torch.uint8, device='cuda'); x.fill_(1); torch.cuda.synchronize(); del x
    small('a 26 GB tensor filled and freed to the cache (no env)')
    torch.cuda.empty_cache()
    small('after torch.cuda.empty_cache()')
    big_env()
    torch.cuda.empty_cache()
    small('a closed 26 GB env, cache emptied at once')
    big_env()
    small('a closed 26 GB env again, cache kept')
