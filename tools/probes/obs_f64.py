#!/usr/bin/env python3
"""obs_dtype='float64' at 4096 x 512: the expansion kernel writing float64 itself (d2d_set_obs_dtype) against the float32 block
cast by the caller afterwards (round 3's VecD2DEnv did `obs.double()` per step)."""
import json
import statistics
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import torch

from gym_d2d_amd.envs import VecD2DEnv

acts = None
for name, cfg, cast in (('float64 written by obs_expand_f64_kernel', {'obs_dtype': 'float64'}, False), ('float32 block + obs.double() per step', {}, True)):
    env = VecD2DEnv(dict(num_rbs=256, num_cues=256, num_due_pairs=256, **cfg), num_envs=4096)
    env.reset(seed=1)
    if acts is None:
        acts = torch.randint(0, 256 * 21, (8, 4096, 512), device=env.device, dtype=torch.int32)
    h = env.simulator.handle
    res = []
    for rnd in range(3):
        for k in range(3):
            o = env.step(acts[k % 8])[0]
            if cast:
                o = o.double()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(10):
            o = env.step(acts[k % 8])[0]
            if cast:
                o = o.double()
        torch.cuda.synchronize()
        res.append((time.perf_counter() - t0) / 10 * 1e3)
    h.profile_reset(); h.profile_enable(True)
    for k in range(10):
        env.step(acts[k % 8])
    ms, n = h.profile_read(1)
    h.profile_enable(False)
    out_bytes = 4096 * 512 * 3072 * (8 if not cast else 4)
    print(json.dumps({'obs': name, 'ms_per_step': round(statistics.median(res), 3), 'expansion_kernel_ms': round(ms / n, 3),
                      'expansion_kernel_GBps': round((out_bytes + 4096 * 512 * 24) / (ms / n) / 1e6, 1)}), flush=True)
    env.close()
    del env, o
    torch.cuda.empty_cache()
