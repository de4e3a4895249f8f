#!/usr/bin/env python3
"""Sweep of the step kernel's threads-per-env and of the interference path (bitmask buckets vs all-pairs), compact
obs mode, interleaved rounds in one process."""
import statistics
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from gym_d2d_amd import _native
from gym_d2d_amd.envs import VecD2DEnv
from gym_d2d_amd.envs.obs_fn import OwnLinkObsFunction


def main():
    b, c, p, r = 4096, 256, 256, 256
    if len(sys.argv) > 1:
        b, c, p, r = map(int, sys.argv[1:5])
    n = c + p
    env = VecD2DEnv({'num_rbs': r, 'num_cues': c, 'num_due_pairs': p, 'obs_fn': OwnLinkObsFunction}, num_envs=b)
    env.reset(seed=1)
    h = env.simulator.handle
    act = env.action_buffer()
    threads = sorted({t for t in (64, 128, 256, 512, 1024) if t <= max(64, ((n + 63) // 64) * 64)})
    variants = [(t, bk) for t in threads for bk in (1, 0)]
    times = {v: [] for v in variants}
    for rnd in range(7):
        for v in variants:
            h.set_tuning(_native.TUNE_STEP_THREADS, v[0])
            h.set_bucketing(bool(v[1]))
            h.profile_reset(); h.profile_enable(True)
            for _ in range(10):
                h.step(act.data_ptr())
            ms, k = h.profile_read(0)
            h.profile_enable(False)
            times[v].append(ms / k * 1e3)
    for med, v, mn in sorted((statistics.median(t), v, min(t)) for v, t in times.items()):
        print(f'threads={v[0]:5d} bucketed={v[1]}  median {med:8.1f} us  min {mn:8.1f} us  -> {b * n / med / 1e3:.2f} G agent-steps/s')
    env.close()


if __name__ == '__main__':
    main()
