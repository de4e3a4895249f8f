#!/usr/bin/env python3
"""Where the small-N fused step kernel's time goes (BASELINE config 2: 1024 envs x 50 links, traffic-model CUEs, LinearObs
expansion fused into the step launch).  Diagnostic build only (D2D_BUILD_DIAG=1 python -m gym_d2d_amd.build --force): lane 0
of every wave stamps s_memtime at the phase boundaries; this reads the stamps of ONE launch.

    python tools/phase_times_small.py [--out profiles/rN_phase_times_default.json]
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

PHASES = ['entry -> loads issued + mask clear', 'barrier 1 wait', 'load wait + decode + stage + mask build', 'barrier 2 wait',
          'walk', 'own link + SINR math + result stores + table to LDS', 'reward reduction + barrier', 'LinearObs expansion (issue of all obs stores)']


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--out', default='')
    args = ap.parse_args()
    b, c, p, r = 1024, 25, 25, 25
    env = VecD2DEnv({'num_rbs': r, 'num_cues': c, 'num_due_pairs': p}, num_envs=b, cue_actions='traffic')
    env.reset(seed=1)
    h = env.simulator.handle
    act = torch.randint(0, r * 21, (64, b, p), device=env.device, dtype=torch.int32)

    def timed():
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        h.step(act[0].data_ptr())
        e0.record()
        for k in range(32):
            h.step(act[k % 64].data_ptr())
        e1.record(); e1.synchronize()
        return e0.elapsed_time(e1) / 32 * 1e3
    h.set_tuning(_native.TUNE_STEP_ABLATE, 0)
    plain_us = float(np.median([timed() for _ in range(9)]))
    # the same launch with the table-only obs mode: the step phase alone as its own kernel
    h.set_obs_mode(_native.OBS_TABLE)
    table_us = float(np.median([timed() for _ in range(9)]))
    h.set_obs_mode(_native.OBS_LINEAR)
    h.set_tuning(_native.TUNE_STEP_ABLATE, 8192)
    for k in range(6):
        h.step(act[k].data_ptr())
    torch.cuda.synchronize()
    waves_per_wg, wgs = 4, b // 4
    raw = np.zeros((b, 16, 16), dtype=np.uint64)
    lib = h._lib
    lib.d2d_debug_stamps.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    lib.d2d_debug_stamps.restype = C.c_int
    assert lib.d2d_debug_stamps(h._h, raw.ctypes.data_as(C.c_void_p), raw.nbytes) == 0
    st = raw.reshape(-1, 16)[: wgs * waves_per_wg, :9].reshape(wgs, waves_per_wg, 9).astype(np.int64)
    for x in range(8):                                         # per-XCD clocks: workgroup g runs on XCD g % 8
        st[x::8] -= st[x::8, :, 0].min()
    d = np.diff(st, axis=2).reshape(-1, 8)
    life = (st[:, :, 8] - st[:, :, 0]).reshape(-1)
    span = int(max(st[x::8, :, 8].max() for x in range(8)))
    out = {'kernel': 'step_kernel<0,1,false,2> 1024 envs x 50 links, 4 envs per 256-thread workgroup, LinearObs expansion fused',
           'launch_us_group_timed': round(plain_us, 2), 'same_launch_table_obs_only_us': round(table_us, 2),
           'obs_bytes': b * 50 * 300 * 4, 'launch_span_ticks': span, 'ticks_per_us_if_span_equals_launch': round(span / plain_us, 1),
           'first_stamp_spread_ticks': {str(q): float(np.percentile(st[:, :, 0], q)) for q in (0, 50, 90, 100)},
           'wave_life_ticks': {'mean': float(life.mean()), 'median': float(np.median(life)), 'p90': float(np.percentile(life, 90))},
           'phases': [{'phase': name, 'mean': round(float(d[:, k].mean()), 1), 'median': float(np.median(d[:, k])),
                       'p90': float(np.percentile(d[:, k], 90)), 'share_of_wave_life': round(float(d[:, k].mean() / life.mean()), 3)}
                      for k, name in enumerate(PHASES)]}
    t_obs = d[:, 7].mean() / (span / plain_us)                 # us spent issuing the expansion
    out['expansion_us_by_the_span_calibration'] = round(float(t_obs), 2)
    out['expansion_GBps'] = round(out['obs_bytes'] / max(t_obs, 1e-9) / 1e3, 1)
    print(json.dumps(out, indent=1))
    if args.out:
        Path(args.out).write_text(json.dumps(out, indent=1))
    env.close()


if __name__ == '__main__':
    main()
