#!/usr/bin/env python3
"""A/B of the obs-expansion kernel variants (LDS-staged vs direct-from-global) around the tuned geometry.
Interleaved rounds in one process; prints median/min HIP-event time per variant."""
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
    variants = [(var, rows, blk) for var in (0, 1) for rows in (1, 2, 3, 4) for blk in (512, 768, 1024)]
    times = {v: [] for v in variants}
    bytes_per = b * n * (24.0 * n + 24.0)
    for rnd in range(8):
        for v in variants:
            h.set_tuning(_native.TUNE_OBS_VARIANT, v[0])
            h.set_tuning(_native.TUNE_OBS_ROWS_PER_WG, v[1])
            h.set_tuning(_native.TUNE_OBS_BLOCK, v[2])
            h.profile_reset(); h.profile_enable(True)
            for _ in range(4):
                h.step(act.data_ptr())
            ms, k = h.profile_read(1)
            h.profile_enable(False)
            times[v].append(ms / k)
    for med, v, mn in sorted((statistics.median(t), v, min(t)) for v, t in times.items()):
        print(f'variant={v[0]} rows={v[1]} block={v[2]:4d}  median {med:.3f} ms  min {mn:.3f} ms  -> {bytes_per / med / 1e6:.0f} GB/s')
    env.close()


if __name__ == '__main__':
    main()
