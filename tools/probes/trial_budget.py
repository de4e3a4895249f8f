#!/usr/bin/env python3
"""How many candidate blocks are worth trying?  In the process states where VecD2DEnv's 15 placement trials found nothing fast (a fresh
process that built a config-2 env right away; bench.py's own process), time 40 candidates allocated by the same scheme (paddings of
MBs, GiB jumps at every third candidate from the seventh on, all held) - no early stop - and print every one."""
import json
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
import torch

from gym_d2d_amd import _native
from gym_d2d_amd.envs import VecD2DEnv

state = sys.argv[1] if len(sys.argv) > 1 else 'fresh'
if state == 'after_headline':
    big = VecD2DEnv({'num_rbs': 256, 'num_cues': 256, 'num_due_pairs': 256}, num_envs=4096)
    big.reset(seed=1)
    a = torch.randint(0, 256 * 21, (4096, 512), device=big.device, dtype=torch.int32)
    for k in range(12):
        big.step(a)
    torch.cuda.synchronize()
    big.close()
env = VecD2DEnv({'num_rbs': 25, 'num_cues': 25, 'num_due_pairs': 25}, num_envs=1024, cue_actions='traffic', placement_trials=0)
env.reset(seed=1234)
h = env.simulator.handle
first = env._t['obs']
nbytes = first.numel() * 4


def timed(n):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        h.step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


for _ in range(30):
    timed(64)
cands, pads, times, jumps = [first], [], [], []
for k in range(40):
    if k:
        pad = ((k * 7) % 11 + 1) * (2 << 20) + (k % 3) * 4096
        if k >= 6 and k % 3 == 0:
            jump = (1 << 30) * (1 + 2 * (((k - 6) // 3) % 6))
            if torch.cuda.mem_get_info()[0] > 4 * jump + nbytes:
                pad = jump
        pads.append(torch.empty(pad, dtype=torch.uint8, device=env.device)); jumps.append(round(pad / 2**20))
        cands.append(torch.empty_like(first))
    h.bind_buffer(_native.BUF_OBS, cands[k].data_ptr(), nbytes)
    timed(32)
    times.append(round(timed(256), 2))
print(json.dumps({'state': state, 'us_per_step': times, 'padding_MiB_before_candidate': [0] + jumps,
                  'classes': ''.join('F' if t < 13.6 else ('m' if t < 14.3 else 's') for t in times)}), flush=True)
env.close()
