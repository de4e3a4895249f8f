#!/usr/bin/env python3
"""Same box, same history (a 26 GB env stepped and closed, its blocks in torch's cache), long steady runs: BASELINE config 2
with its tensors from the default caching allocator, from a private torch MemPool (obs only / everything)."""
import json
import os
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import torch

from gym_d2d_amd.envs import VecD2DEnv

big = VecD2DEnv({'num_rbs': 256, 'num_cues': 256, 'num_due_pairs': 256}, num_envs=4096)
big.reset(seed=1)
a = torch.randint(0, 256 * 21, (4096, 512), device=big.device, dtype=torch.int32)
for k in range(5):
    big.step(a)
torch.cuda.synchronize()
big.close()
del big, a
for rnd in range(3):
    for policy in ('none', 'obs', 'all'):
        os.environ['D2D_VEC_ENV_POOL'] = policy
        env = VecD2DEnv({'num_rbs': 25, 'num_cues': 25, 'num_due_pairs': 25}, num_envs=1024, cue_actions='traffic')
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
        print(json.dumps({'round': rnd, 'pool': policy, 'us_per_step': res, 'mem_reserved_GB': round(torch.cuda.memory_reserved() / 1e9, 2)}), flush=True)
        env.close()
        del env, acts
