#!/usr/bin/env python3
"""Wall-clock latency of the drop-in single-env D2DEnv.step (host dicts in / out), with a cProfile breakdown."""
import cProfile
import pstats
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np

from gym_d2d_amd.envs import D2DEnv


def run(c, p, r, steps, profile):
    env = D2DEnv({'num_rbs': r, 'num_cues': c, 'num_due_pairs': p})
    obs = env.reset()
    rng = np.random.default_rng(0)
    acts = [{k: int(rng.integers(0, env.action_space['due' if k.startswith('due') else 'cue'].n)) for k in obs} for _ in range(8)]
    for k in range(10):
        env.step(acts[k % 8])
    t0 = time.perf_counter()
    for k in range(steps):
        env.step(acts[k % 8])
    ms = (time.perf_counter() - t0) / steps * 1e3
    print(f'{c}/{p}/{r}: {ms:.3f} ms per env.step')
    if profile:
        pr = cProfile.Profile()
        pr.enable()
        for k in range(steps):
            env.step(acts[k % 8])
        pr.disable()
        pstats.Stats(pr).sort_stats('cumulative').print_stats(14)
    env.close()


if __name__ == '__main__':
    prof = '--profile' in sys.argv
    run(25, 25, 25, 300, prof)
    run(256, 256, 256, 100, prof)
