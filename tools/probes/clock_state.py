#!/usr/bin/env python3
"""Is the small workload's 13.3 <-> 15.2 us per step the chip's clock / power state rather than where its buffers sit?
A fresh process times BASELINE config 2 in consecutive chunks; then right behind a heavy burst (a 26 GB env stepping);
then after an idle pause; then continuously for a second."""
import json
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import torch

from gym_d2d_amd.envs import VecD2DEnv


def chunks(env, acts, n_chunks, steps, tag):
    out = []
    for c in range(n_chunks):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(steps):
            env.step(acts[k % 8])
        torch.cuda.synchronize()
        out.append(round((time.perf_counter() - t0) / steps * 1e6, 2))
    print(json.dumps({'when': tag, 'steps_per_chunk': steps, 'us_per_step_by_chunk': out}), flush=True)


env = VecD2DEnv({'num_rbs': 25, 'num_cues': 25, 'num_due_pairs': 25}, num_envs=1024, cue_actions='traffic')
env.reset(seed=1)
acts = torch.randint(0, 25 * 21, (8, 1024, 25), device=env.device, dtype=torch.int32)
chunks(env, acts, 8, 400, 'fresh process, first 3200 steps')
big = VecD2DEnv({'num_rbs': 256, 'num_cues': 256, 'num_due_pairs': 256}, num_envs=4096)
big.reset(seed=1)
a = torch.randint(0, 256 * 21, (4096, 512), device=big.device, dtype=torch.int32)
chunks(env, acts, 4, 400, 'a 26 GB env allocated and reset (one 3.6 ms burst), not stepping')
for k in range(30):
    big.step(a)
chunks(env, acts, 8, 400, 'right behind 30 steps of the 26 GB env (110 ms of HBM-bound load)')
time.sleep(1.0)
chunks(env, acts, 4, 400, 'after 1 s idle')
for k in range(30):
    big.step(a)
big.close()
del big, a
torch.cuda.empty_cache()
chunks(env, acts, 8, 400, 'behind 30 big steps, big env closed and its memory returned')
time.sleep(1.0)
chunks(env, acts, 4, 400, 'after 1 s idle, big env gone')
chunks(env, acts, 10, 7000, 'continuous: 10 chunks of 7000 steps (about 1 s)')
env.close()
