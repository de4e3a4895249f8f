"""Thousands of environments per step: the batched API (no counterpart in the reference)."""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))      # run from a checkout without installing

import torch

from gym_d2d_amd.envs import VecD2DEnv

env = VecD2DEnv({'num_rbs': 25, 'num_cues': 25, 'num_due_pairs': 25}, num_envs=1024, cue_actions='traffic')
obs = env.reset(seed=1234)                             # [1024, 50, 300] float32 on the GPU
for _ in range(10):
    due_actions = torch.randint(0, 25 * 21, (1024, 25), device=obs.device, dtype=torch.int32)
    obs, rewards, dones, info = env.step(due_actions)  # CUEs follow the UplinkTrafficModel
print('mean system-capacity reward:', float(rewards[:, 0].mean()), '| done:', bool(dones.all()))
