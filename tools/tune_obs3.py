#!/usr/bin/env python3
"""Obs expansion: how many envs should one XCD write concurrently (xcd_remap = G)?"""
import statistics
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from gym_d2d_amd import _native
from gym_d2d_amd.envs import VecD2DEnv


def main():
    b, c, p, r = 4096, 256, 256, 256
    n = c + p
    env = VecD2DEnv({'num_rbs': r, 'num_cues': c, 'num_due_pairs': p}, num_envs=b)
    env.reset(seed=1)
    h = env.simulator.handle
    act = env.action_buffer()
    variants = [(g, rows, blk) for g in (1, 2, 4, 8, 16) for rows, blk in ((2, 768), (1, 512), (2, 1024), (4, 768))]
    times = {v: [] for v in variants}
    bytes_per = b * n * (24.0 * n + 24.0)
    for rnd in range(8):
        for v in variants:
            h.set_tuning(_native.TUNE_OBS_XCD_REMAP, v[0])
            h.set_tuning(_native.TUNE_OBS_ROWS_PER_WG, v[1])
            h.set_tuning(_native.TUNE_OBS_BLOCK, v[2])
            h.profile_reset(); h.profile_enable(True)
            for _ in range(4):
                h.step(act.data_ptr())
            ms, k = h.profile_read(1)
            h.profile_enable(False)
            times[v].append(ms / k)
    for med, v, mn in sorted((statistics.median(t), v, min(t)) for v, t in times.items()):
        print(f'G={v[0]:2d} rows={v[1]} block={v[2]:4d}  median {med:.3f} ms  min {mn:.3f} ms  -> {bytes_per / med / 1e6:.0f} GB/s')
    env.close()


if __name__ == '__main__':
    main()
