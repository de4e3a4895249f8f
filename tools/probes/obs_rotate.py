#!/usr/bin/env python3
"""The fused small-N step (1024 x 50, LinearObs) across ALLOCATIONS (a fresh set of buffers per round: where they land physically
moves the time per step by 15 %) x start-phase multipliers of the fused expansion (D2D_TUNE_STEP_OBS_ROTATE; 0 = every
workgroup walks its region from the start)."""
import json
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import torch

from gym_d2d_amd import _native
from gym_d2d_amd.envs import VecD2DEnv

ROT = [0, 29, 1, 7, 17, 37, 53]
hold = []
for alloc in range(int(sys.argv[1]) if len(sys.argv) > 1 else 8):
    env = VecD2DEnv({'num_rbs': 25, 'num_cues': 25, 'num_due_pairs': 25}, num_envs=1024, cue_actions='traffic')
    env.reset(seed=1)
    h = env.simulator.handle
    acts = torch.randint(0, 25 * 21, (220, 1024, 25), device=env.device, dtype=torch.int32)
    row = {}
    for rnd in range(2):
        for rot in ROT:
            h.set_tuning(_native.TUNE_STEP_OBS_ROTATE, rot)
            for k in range(20):
                env.step(acts[k])
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for k in range(20, 220):
                env.step(acts[k])
            torch.cuda.synchronize()
            row.setdefault(rot, []).append(round((time.perf_counter() - t0) / 200 * 1e6, 2))
    print(json.dumps({'allocation': alloc, 'us_per_step_by_rotate': {str(k): min(v) for k, v in row.items()}}), flush=True)
    env.close()
    del env, acts
    # perturb where the next allocation lands
    if alloc % 2 == 0:
        torch.cuda.empty_cache()
    else:
        hold.append(torch.empty(int(3e8) * (alloc + 1), dtype=torch.uint8, device='cuda'))
