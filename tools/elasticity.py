#!/usr/bin/env python3
"""Which pipe does the step kernel's time hang on?  Diagnostic build (D2D_BUILD_DIAG=1 python -m gym_d2d_amd.build --force):
the HOT kernel at 4096 x 512 with a measured dose of extra work of ONE kind per wave, interleaved with the plain kernel.

    python tools/elasticity.py [--out profiles/rN_elasticity_step_kernel.json]
"""
import argparse
import json
import statistics
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
sys.path.insert(0, str(Path(__file__).resolve().parent))
import torch

from ab_step import timed
from gym_d2d_amd import _native
from gym_d2d_amd.envs import VecD2DEnv
from gym_d2d_amd.envs.obs_fn import OwnLinkObsFunction

DOSES = {0: 'plain', 16384: '+64 VALU (two FMA chains)', 32768: '+64 SALU + 32 scalar moves', 65536: '+8 random 16-byte LDS reads per lane',
         131072: '+8 LDS atomics per lane (random words)', 262144: '+4 global 4-byte stores per lane'}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--out', default='')
    args = ap.parse_args()
    b, c, p, r = 4096, 256, 256, 256
    env = VecD2DEnv({'num_rbs': r, 'num_cues': c, 'num_due_pairs': p, 'obs_fn': OwnLinkObsFunction}, num_envs=b)
    env.reset(seed=1)
    h = env.simulator.handle
    act = torch.randint(0, r * 21, (64, b, c + p), device=env.device, dtype=torch.int32)
    times = {d: [] for d in DOSES}
    for rnd in range(9):
        for d in DOSES:
            h.set_tuning(_native.TUNE_STEP_ABLATE, d)
            times[d].append(timed(h, act, 32))
    base = statistics.median(times[0])
    out = [{'dose': DOSES[d], 'median_us': round(statistics.median(times[d]), 2), 'delta_us': round(statistics.median(times[d]) - base, 2)}
           for d in DOSES]
    for o in out:
        print(json.dumps(o))
    if args.out:
        Path(args.out).write_text(json.dumps({'kernel': 'step_kernel<0,1,true,HOT> 4096 x 512, compact-obs mode, SystemCapacity', 'doses': out}, indent=1))
    env.close()


if __name__ == '__main__':
    main()
