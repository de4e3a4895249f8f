#!/usr/bin/env python3
"""A/B sweep of the obs-expansion kernel's launch knobs, interleaved rounds in ONE process (guide rule 24).
Prints median / min HIP-event time per variant and the write-only fill probe (on-box ceiling)."""
import itertools
import json
import statistics
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch

from gym_d2d_amd import _native
from gym_d2d_amd.envs import VecD2DEnv


def main():
    b, c, p, r = 4096, 256, 256, 256
    if len(sys.argv) > 1:
        b, c, p, r = map(int, sys.argv[1:5])
    n = c + p
    env = VecD2DEnv({'num_rbs': r, 'num_cues': c, 'num_due_pairs': p}, num_envs=b)
    env.reset(seed=1)
    h = env.simulator.handle
    act = env.action_buffer()
    probe = [h.probe_write_bandwidth(8 << 30, 5) for _ in range(3)]
    print('fill probe GB/s (8 GiB, nontemporal 16B/lane):', [round(x) for x in probe])
    rows_opts = [0, 1, 2, 3, 4, 6] if n == 512 else [0, 1, 2, 5, 10, 25, 50]
    variants = [(rows, nt, blk, xcd) for rows in rows_opts for blk in (384, 512, 640, 768, 896, 1024) for xcd in (1, 0) for nt in (1, 0)
                if rows <= n and (xcd == 1 or nt == 1)]
    times = {v: [] for v in variants}
    bytes_per = b * n * (24.0 * n + 24.0)
    for rnd in range(5):
        for v in variants:
            h.set_tuning(_native.TUNE_OBS_ROWS_PER_WG, v[0])
            h.set_tuning(_native.TUNE_OBS_NONTEMPORAL, v[1])
            h.set_tuning(_native.TUNE_OBS_BLOCK, v[2])
            h.set_tuning(_native.TUNE_OBS_XCD_REMAP, v[3])
            h.profile_reset(); h.profile_enable(True)
            for _ in range(3):
                h.step(act.data_ptr())
            ms, k = h.profile_read(1)
            h.profile_enable(False)
            times[v].append(ms / k)
    out = []
    for v in variants:
        med, mn = statistics.median(times[v]), min(times[v])
        out.append((med, v, mn))
    for med, v, mn in sorted(out):
        print(f'rows={v[0]:4d} nt={v[1]} block={v[2]:4d} xcd={v[3]}  median {med:.3f} ms  min {mn:.3f} ms  -> {bytes_per / med / 1e6:.0f} GB/s')
    env.close()


if __name__ == '__main__':
    main()
