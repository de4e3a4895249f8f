#!/usr/bin/env python3
"""Obs-expansion geometry sweep for the small default workload (B x 50 links): rows per workgroup x block size."""
import statistics
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from gym_d2d_amd import _native
from gym_d2d_amd.envs import VecD2DEnv


def main():
    b, c, p, r = 1024, 25, 25, 25
    if len(sys.argv) > 1:
        b, c, p, r = map(int, sys.argv[1:5])
    n = c + p
    env = VecD2DEnv({'num_rbs': r, 'num_cues': c, 'num_due_pairs': p}, num_envs=b)
    env.reset(seed=1)
    h = env.simulator.handle
    act = env.action_buffer()
    variants = [(0, 0)] + [(rows, blk) for rows in (1, 2, 3, 5, 7, 10, 13, 25, 50) for blk in (64, 128, 256, 512) if rows <= n]
    times = {v: [] for v in variants}
    bytes_per = b * n * (24.0 * n + 24.0)
    for rnd in range(7):
        for v in variants:
            h.set_tuning(_native.TUNE_OBS_ROWS_PER_WG, v[0])
            h.set_tuning(_native.TUNE_OBS_BLOCK, v[1])
            h.profile_reset(); h.profile_enable(True)
            for _ in range(20):
                h.step(act.data_ptr())
            ms, k = h.profile_read(1)
            h.profile_enable(False)
            times[v].append(ms / k * 1e3)
    for med, v in sorted((statistics.median(t), v) for v, t in times.items())[:12]:
        print(f'rows={v[0]:3d} block={v[1]:4d}  median {med:7.2f} us -> {bytes_per / med / 1e3:.0f} GB/s')
    print('auto:', statistics.median(times[(0, 0)]))
    env.close()


if __name__ == '__main__':
    main()
