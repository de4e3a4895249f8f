#!/usr/bin/env python3
"""What is the rollout kernel's same-RB walk DIVERGENCE worth?  The same 4096 x 512 step with i.i.d. random actions (interferers per
link ~ Poisson(2): a wave runs the maximum of its 64 lanes) against action patterns with the same or less total pair work and NO
imbalance: exactly 3 links on every used RB (2 interferers per link, like the random mean), exactly 2 (1 interferer), every link
alone on its RB ... (needs R >= N: not available at 256 RBs, so 'alone' is emulated by 2 per RB with the partner far away - same
work as 2 per RB).  Upper bound for what any load-balanced walk could save."""
import json
import statistics
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch

from ab_step import timed
from gym_d2d_amd.envs import VecD2DEnv
from gym_d2d_amd.envs.obs_fn import OwnLinkObsFunction, SignalPlanesObsFunction

b, c, p, r = 4096, 256, 256, 256
for mode in ('none', 'table'):
    if mode == 'none':
        env = VecD2DEnv({'num_rbs': r, 'num_cues': c, 'num_due_pairs': p, 'obs_fn': SignalPlanesObsFunction}, num_envs=b, export_actions=False,
                        reward_per_env=True, placement_trials=0)
    else:
        env = VecD2DEnv({'num_rbs': r, 'num_cues': c, 'num_due_pairs': p, 'obs_fn': OwnLinkObsFunction}, num_envs=b, export_actions=False,
                        placement_trials=0)
    env.reset(seed=1)
    h = env.simulator.handle
    dev = env.device
    pc, pd = env.num_pwr_actions['cue'], env.num_pwr_actions['due']
    levels = torch.tensor([pc] * c + [pd] * p, device=dev, dtype=torch.int32)

    def pattern(per_rb):
        out = torch.empty((64, b, c + p), dtype=torch.int32, device=dev)
        for k in range(64):
            perm = torch.argsort(torch.rand(b, c + p, device=dev), dim=1)             # a random permutation of the links per env
            rb = torch.empty_like(perm)
            rb.scatter_(1, perm, (torch.arange(c + p, device=dev) // per_rb).expand(b, -1))
            out[k] = (rb.to(torch.int32) * levels + torch.randint(0, 20, (b, c + p), device=dev, dtype=torch.int32))
        return out
    acts = {'i.i.d. uniform RBs (Poisson(2) interferers per link)': torch.randint(0, r * 21, (64, b, c + p), device=dev, dtype=torch.int32),
            'exactly 3 links per used RB (2 interferers per link, balanced)': pattern(3),
            'exactly 2 links per RB (1 interferer per link, balanced)': pattern(2),
            'exactly 4 links per used RB (3 interferers per link, balanced)': pattern(4)}
    for k in range(1500):
        h.step(acts['i.i.d. uniform RBs (Poisson(2) interferers per link)'][k % 64].data_ptr())
    times = {n: [] for n in acts}
    for rnd in range(7):
        for n, a in acts.items():
            timed(h, a, 64)
            times[n].append(timed(h, a, 256))
    for n, t in times.items():
        print(json.dumps({'obs_mode': mode, 'actions': n, 'median_us': round(statistics.median(t), 2), 'min_us': round(min(t), 2)}), flush=True)
    env.close()
