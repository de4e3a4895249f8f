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

DOSES = {0: 'plain', 524288: '+64 full-rate VALU (v_xor / v_add_u32, four independent chains)', 16384: '+32 v_pk_fma_f32 (two dependent FMA chains, packed by the compiler)', 32768: '+64 SALU + 32 scalar moves', 65536: '+8 random 16-byte LDS reads per lane',
         131072: '+8 LDS atomics per lane (random words)', 262144: '+4 global 4-byte stores per lane'}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--out', default='')
    ap.add_argument('--mode', default='table', choices=['table', 'none'], help="none = the obs-less learner configuration (36 bytes per link)")
    args = ap.parse_args()
    b, c, p, r = 4096, 256, 256, 256
    if args.mode == 'none':
        from gym_d2d_amd.envs.obs_fn import SignalPlanesObsFunction
        env = VecD2DEnv({'num_rbs': r, 'num_cues': c, 'num_due_pairs': p, 'obs_fn': SignalPlanesObsFunction}, num_envs=b, export_actions=False,
                        reward_per_env=True, placement_trials=0)
    else:
        env = VecD2DEnv({'num_rbs': r, 'num_cues': c, 'num_due_pairs': p, 'obs_fn': OwnLinkObsFunction}, num_envs=b, placement_trials=0)
    env.reset(seed=1)
    h = env.simulator.handle
    act = torch.randint(0, r * 21, (64, b, c + p), device=env.device, dtype=torch.int32)
    times = {d: [] for d in DOSES}
    for k in range(1500):                      # past the clock ramp
        h.step(act[k % 64].data_ptr())
    for rnd in range(9):
        for d in DOSES:
            h.set_tuning(_native.TUNE_STEP_ABLATE, d)
            timed(h, act, 64)
            times[d].append(timed(h, act, 256))
    base = statistics.median(times[0])
    out = [{'dose': DOSES[d], 'median_us': round(statistics.median(times[d]), 2), 'delta_us': round(statistics.median(times[d]) - base, 2)}
           for d in DOSES]
    for o in out:
        print(json.dumps(o))
    if args.out:
        Path(args.out).write_text(json.dumps({'kernel': f'step_kernel<0,1,true,HOT> 4096 x 512, obs mode {args.mode}, SystemCapacity', 'doses': out}, indent=1))
    env.close()


if __name__ == '__main__':
    main()
