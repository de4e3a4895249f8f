#!/usr/bin/env python3
"""Which cache policy drains the obs stream fastest?  The 16-byte stores of the obs kernel (and of the fill family) under
plain / nt / sc1 / sc0 sc1 / sc0 sc1 nt / sc1 nt, interleaved rounds in one process, on the real 25.8 GB obs block."""
import json
import statistics
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import torch

from gym_d2d_amd import _native
from gym_d2d_amd.envs import VecD2DEnv

NAMES = {0: 'plain', 1: 'nt', 2: 'sc1', 3: 'sc0 sc1', 4: 'sc0 sc1 nt', 5: 'sc1 nt'}
env = VecD2DEnv({'num_rbs': 256, 'num_cues': 256, 'num_due_pairs': 256}, num_envs=4096)
env.reset(seed=1)
h = env.simulator.handle
act = torch.randint(0, 256 * 21, (8, 4096, 512), device=env.device, dtype=torch.int32)
times = {p: [] for p in NAMES}
for rnd in range(6):
    for p in NAMES:
        h.set_tuning(_native.TUNE_OBS_NONTEMPORAL, p)
        for k in range(3):
            h.step(act[k % 8].data_ptr())
        h.profile_reset(); h.profile_enable(True)
        for k in range(12):
            h.step(act[k % 8].data_ptr())
        ms, n = h.profile_read(1)
        h.profile_enable(False)
        times[p].append(ms / n)
h.set_tuning(_native.TUNE_OBS_NONTEMPORAL, 1)
bytes_per_launch = 4096 * 512 * (24 * 512 + 24)
for p, t in times.items():
    print(json.dumps({'kernel': 'obs_expand_kernel', 'store_policy': NAMES[p], 'median_ms': round(statistics.median(t), 4), 'min_ms': round(min(t), 4),
                      'median_GBps': round(bytes_per_launch / statistics.median(t) / 1e6, 1)}), flush=True)
obs_ptr, obs_bytes = h.get_buffer(_native.BUF_OBS)
nbytes = (obs_bytes // (64 << 20)) * (64 << 20)
torch.cuda.synchronize()
for geom, gname in ((0, '768 threads x 2 rows'), (1, '1024 threads x 2 rows')):
    for stage, sname in ((0, 'plain fill'), (32, 'LDS stage + barrier')):
        for pol, pname in ((16, 'plain'), (0, 'nt'), (128, 'sc1'), (256, 'sc0 sc1'), (384, 'sc0 sc1 nt'), (512, 'sc1 nt')):
            r = [h.probe_write_staged(nbytes, geom + stage + pol, 0, iters=3, dst_ptr=obs_ptr) for _ in range(3)]
            print(json.dumps({'fill_over': 'the obs block', 'geometry': gname, 'form': sname, 'store_policy': pname,
                              'GBps_median': round(statistics.median(r), 1), 'GBps_best': round(max(r), 1)}), flush=True)
env.close()
