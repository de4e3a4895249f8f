#!/usr/bin/env python3
"""Does what ran before in the process change the small workload's time per step?  (bench.py's other_workloads.default read
17.2 us where the stand-alone `--workload default` run on the same box read 15.0 us.)"""
import json
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import torch

from gym_d2d_amd.envs import VecD2DEnv
from gym_d2d_amd.envs.obs_fn import LinearObsFunction


def small(tag):
    env = VecD2DEnv({'num_rbs': 25, 'num_cues': 25, 'num_due_pairs': 25}, num_envs=1024, cue_actions='traffic')
    env.reset(seed=1)
    acts = torch.randint(0, 25 * 21, (220, 1024, 25), device=env.device, dtype=torch.int32)
    res = []
    for rnd in range(3):
        for k in range(20):
            env.step(acts[k])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(20, 220):
            env.step(acts[k])
        torch.cuda.synchronize()
        res.append(round((time.perf_counter() - t0) / 200 * 1e6, 2))
    print(json.dumps({'after': tag, 'us_per_step': res, 'mem_reserved_GB': round(torch.cuda.memory_reserved() / 1e9, 2)}), flush=True)
    env.close()
    del env, acts


small('nothing (fresh process)')
small('a small env before it')
big = VecD2DEnv({'num_rbs': 256, 'num_cues': 256, 'num_due_pairs': 256}, num_envs=4096)
big.reset(seed=1)
a = torch.randint(0, 256 * 21, (4096, 512), device=big.device, dtype=torch.int32)
for k in range(30):
    big.step(a)
torch.cuda.synchronize()
small('a live 26 GB stress env that just stepped 30 times')
big.close()
del big
small('the stress env closed (its tensors in torch\'s cache)')
torch.cuda.empty_cache()
small('torch.cuda.empty_cache()')
x = torch.empty(int(60e9), dtype=torch.uint8, device='cuda')
x.fill_(1)
torch.cuda.synchronize()
del x
small('a 60 GB fill, tensor freed to the cache')
torch.cuda.empty_cache()
small('empty_cache again')
