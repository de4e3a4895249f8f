#!/usr/bin/env python3
"""Where a step-kernel workgroup's life goes (diagnostic build only: D2D_BUILD_DIAG=1 python -m gym_d2d_amd.build --force).

Lane 0 of every wave stamps the shader clock (s_memtime) at the phase boundaries of csrc/d2d_step.hip; this reads the
stamps of ONE launch at 4096 x 512 (compact-obs mode) and prints, per phase, the mean / median / p90 over all waves, the
launch's own timeline (when workgroups start and end) and how many waves sit in each phase at sampled instants.

    python tools/phase_times.py [--out profiles/rN_phase_times.json]
"""
import argparse
import ctypes as C
import json
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch

from gym_d2d_amd import _native
from gym_d2d_amd.envs import VecD2DEnv
from gym_d2d_amd.envs.obs_fn import OwnLinkObsFunction

PHASES = ['entry -> loads issued + mask clear', 'barrier 1 wait', 'load wait + decode + stage + mask build', 'barrier 2 wait',
          'walk', 'own link + SINR math + stores issued', 'reduction / ticket / (last wave) reward row']


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--out', default='')
    ap.add_argument('--reward', type=int, default=1)
    args = ap.parse_args()
    b, c, p, r = 4096, 256, 256, 256
    env = VecD2DEnv({'num_rbs': r, 'num_cues': c, 'num_due_pairs': p, 'obs_fn': OwnLinkObsFunction}, num_envs=b)
    env.reset(seed=1)
    h = env.simulator.handle
    h.set_reward(args.reward, 0.0 if args.reward != 2 else -70.0)
    act = torch.randint(0, r * 21, (8, b, c + p), device=env.device, dtype=torch.int32)
    h.set_tuning(_native.TUNE_STEP_ABLATE, 8192)
    for k in range(6):
        h.step(act[k].data_ptr())
    torch.cuda.synchronize()
    # timed without stamps for the cycle -> us calibration
    h.set_tuning(_native.TUNE_STEP_ABLATE, 0)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    h.step(act[0].data_ptr())
    e0.record()
    for k in range(32):
        h.step(act[k % 8].data_ptr())
    e1.record(); e1.synchronize()
    plain_us = e0.elapsed_time(e1) / 32 * 1e3
    h.set_tuning(_native.TUNE_STEP_ABLATE, 8192)
    h.step(act[7].data_ptr())
    torch.cuda.synchronize()
    waves = (c + p) // 64
    raw = np.zeros((b, 16, 16), dtype=np.uint64)
    lib = h._lib
    lib.d2d_debug_stamps.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    lib.d2d_debug_stamps.restype = C.c_int
    assert lib.d2d_debug_stamps(h._h, raw.ctypes.data_as(C.c_void_p), raw.nbytes) == 0
    st = raw.reshape(-1, 16)[: b * waves, :8].reshape(b, waves, 8).astype(np.int64)
    # every XCD counts on its own base: workgroup g runs on XCD g % 8 (round-robin dispatch) -> normalise per XCD
    for x in range(8):
        st[x::8] -= st[x::8, :, 0].min()
    spans = [int(st[x::8, :, 7].max()) for x in range(8)]
    span = max(spans)
    d = np.diff(st, axis=2).reshape(-1, 7)                       # [waves, 7 phases]
    life = (st[:, :, 7] - st[:, :, 0]).reshape(-1)
    out = {'kernel': 'step_kernel<0,1,true> 4096 x 512, compact-obs mode, reward_fn %d' % args.reward,
           'clock': 's_memtime ticks (per-XCD counters, normalised to the first stamp of each XCD)', 'launch_span_ticks_per_xcd': spans, 'launch_span_ticks': int(span), 'unstamped_launch_us_group_timed': round(plain_us, 2),
           'ticks_per_us_if_span_equals_that': round(span / plain_us, 1),
           'wave_life_ticks': {'mean': float(life.mean()), 'median': float(np.median(life)), 'p90': float(np.percentile(life, 90))},
           'phases': []}
    for k, name in enumerate(PHASES):
        col = d[:, k]
        out['phases'].append({'phase': name, 'mean': round(float(col.mean()), 1), 'median': float(np.median(col)),
                              'p90': float(np.percentile(col, 90)), 'share_of_wave_life': round(float(col.mean() / life.mean()), 3)})
    wg_start = st[:, :, 0].min(axis=1); wg_end = st[:, :, 7].max(axis=1)
    order = np.argsort(wg_start)
    out['workgroup_start_ticks_percentiles'] = {str(q): float(np.percentile(wg_start, q)) for q in (0, 10, 25, 50, 75, 90, 100)}
    out['workgroup_end_ticks_percentiles'] = {str(q): float(np.percentile(wg_end, q)) for q in (0, 10, 25, 50, 75, 90, 100)}
    out['workgroup_life_ticks'] = {'mean': float((wg_end - wg_start).mean()), 'median': float(np.median(wg_end - wg_start))}
    # occupancy by phase at sampled instants: how many waves are inside each phase
    samples = np.linspace(0, span, 21)[1:-1]
    occ = []
    flat = st[0::8].reshape(-1, 8)                             # XCD 0 only: one clock domain
    for t in samples:
        row = [int(((flat[:, k] <= t) & (t < flat[:, k + 1])).sum()) for k in range(7)]
        occ.append({'t': int(t), 'waves_in_phase': row})
    out['occupancy_samples_xcd0'] = occ
    # elasticity to VALU work: 64 extra (independent-of-memory) FMAs per wave, about +16 % VALU instructions
    def timed(ab):
        h.set_tuning(_native.TUNE_STEP_ABLATE, ab)
        h.step(act[0].data_ptr())
        e0.record()
        for k in range(32):
            h.step(act[k % 8].data_ptr())
        e1.record(); e1.synchronize()
        return e0.elapsed_time(e1) / 32 * 1e3
    base, more = [], []
    for _ in range(9):
        base.append(timed(0)); more.append(timed(16384))
    out['valu_elasticity'] = {'launch_us': round(float(np.median(base)), 2), 'launch_us_with_64_more_valu_per_wave': round(float(np.median(more)), 2),
                              'note': 'a VALU-bound kernel would slow down by about 16 %'}
    print(json.dumps(out, indent=1))
    if args.out:
        Path(args.out).write_text(json.dumps(out, indent=1))
    env.close()


if __name__ == '__main__':
    main()
