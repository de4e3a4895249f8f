#!/usr/bin/env python3
"""One allocation history, then K steps of the small workload (BASELINE config 2) - the program rocprofv3 --pmc wraps to tell
WHY the same launch takes 14.9 ... 17.4 us depending on what the process did before (profiles/r3_context_effect_default.jsonl):
address translation (TCP_UTCL1_*) or HBM channel balance (per-instance TCC_EA0_WRREQ / _STALL).

    python3 tools/probes/context_pmc.py --state fresh|big_live|big_freed|big_freed_empty [--steps K] [--time]
"""
import argparse
import json
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import torch

from gym_d2d_amd.envs import VecD2DEnv

ap = argparse.ArgumentParser()
ap.add_argument('--state', default='fresh')
ap.add_argument('--steps', type=int, default=12)
ap.add_argument('--time', action='store_true')
ap.add_argument('--own', default='', help="'obs' / 'all': these buffers are handed back to the library (its own hipMalloc) instead of torch tensors")
ap.add_argument('--rotate', type=int, default=-1)
a = ap.parse_args()

big = None
if a.state != 'fresh':
    big = VecD2DEnv({'num_rbs': 256, 'num_cues': 256, 'num_due_pairs': 256}, num_envs=4096)
    big.reset(seed=1)
    act = torch.randint(0, 256 * 21, (4096, 512), device=big.device, dtype=torch.int32)
    for k in range(3):
        big.step(act)
    torch.cuda.synchronize()
    if a.state in ('big_freed', 'big_freed_empty'):
        big.close()
        del big, act
        big = None
        if a.state == 'big_freed_empty':
            torch.cuda.empty_cache()

env = VecD2DEnv({'num_rbs': 25, 'num_cues': 25, 'num_due_pairs': 25}, num_envs=1024, cue_actions='traffic')
h = env.simulator.handle
from gym_d2d_amd import _native
if a.own:
    which = [_native.BUF_OBS] if a.own == 'obs' else [_native.BUF_OBS, _native.BUF_OBS_TABLE, _native.BUF_SINR_DB, _native.BUF_SNR_DB,
                                                       _native.BUF_RATE_BPS, _native.BUF_CAPACITY, _native.BUF_REWARD, _native.BUF_RB, _native.BUF_PWR]
    for w in which:
        h.bind_buffer(w, 0, 0)                  # unbound: the library allocates its own on the next step
if a.rotate >= 0:
    h.set_tuning(_native.TUNE_STEP_OBS_ROTATE, a.rotate)
env.reset(seed=1)
acts = torch.randint(0, 25 * 21, (8, 1024, 25), device=env.device, dtype=torch.int32)
if a.time:
    res = []
    for rnd in range(3):
        for k in range(20):
            env.step(acts[k % 8])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(200):
            env.step(acts[k % 8])
        torch.cuda.synchronize()
        res.append(round((time.perf_counter() - t0) / 200 * 1e6, 2))
    obs_ptr = h.get_buffer(_native.BUF_OBS)[0]
    print(json.dumps({'state': a.state, 'own': a.own, 'rotate': a.rotate, 'obs_ptr_mod_2MiB': obs_ptr % (2 << 20), 'obs_ptr_GiB': round(obs_ptr / 2**30, 3),
                      'us_per_step': res, 'mem_reserved_GB': round(torch.cuda.memory_reserved() / 1e9, 2)}), flush=True)
else:
    for k in range(a.steps):
        env.step(acts[k % 8])
    torch.cuda.synchronize()
env.close()
