#!/usr/bin/env python3
"""Small-N obs expansion: LDS-staged (variant 0) vs direct-from-global (variant 1) kernels."""
import statistics
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from gym_d2d_amd import _native
from gym_d2d_amd.envs import VecD2DEnv


def main():
    b, c, p, r = map(int, sys.argv[1:5]) if len(sys.argv) > 1 else (1024, 25, 25, 25)
    n = c + p
    env = VecD2DEnv({'num_rbs': r, 'num_cues': c, 'num_due_pairs': p}, num_envs=b)
    env.reset(seed=1)
    h = env.simulator.handle
    act = env.action_buffer()
    variants = [(var, rows, blk) for var in (0, 1) for rows in (2, 5, 7, 10, 25, 50) for blk in (128, 256, 512)]
    times = {v: [] for v in variants}
    bytes_per = b * n * (24.0 * n + 24.0)
    for rnd in range(7):
        for v in variants:
            h.set_tuning(_native.TUNE_OBS_VARIANT, v[0])
            h.set_tuning(_native.TUNE_OBS_ROWS_PER_WG, v[1])
            h.set_tuning(_native.TUNE_OBS_BLOCK, v[2])
            h.profile_reset(); h.profile_enable(True)
            for _ in range(20):
                h.step(act.data_ptr())
            ms, k = h.profile_read(1)
            h.profile_enable(False)
            times[v].append(ms / k * 1e3)
    for med, v in sorted((statistics.median(t), v) for v, t in times.items())[:10]:
        print(f'variant={v[0]} rows={v[1]:3d} block={v[2]:4d}  median {med:7.2f} us -> {bytes_per / med / 1e3:.0f} GB/s')
    env.close()


if __name__ == '__main__':
    main()
