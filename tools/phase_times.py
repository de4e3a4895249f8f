#!/usr/bin/env python3
"""Where a rollout-kernel wave's life goes (diagnostic build only: D2D_BUILD_DIAG=1 python -m gym_d2d_amd.build --force).

Lane 0 of every wave stamps the shader clock (s_memtime) at the phase boundaries of csrc/d2d_rollout.hip; this reads the
stamps of ONE launch at 4096 x 512 and prints, per phase, the mean / median / p90 over all waves, when workgroups start and
end, and how many waves of one XCD sit in each phase at sampled instants.

    python tools/phase_times.py [--mode none|table] [--lpt 1|2] [--out profiles/rN_phase_times_rollout_kernel.json]
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
from gym_d2d_amd.envs.obs_fn import OwnLinkObsFunction, SignalPlanesObsFunction

PHASES = ['entry -> loads issued + pass 0 (slots, counters, flags)', 'barrier 1 wait', 'load wait + pass 1 (decode, tuple, list entry)',
          'barrier 2 wait', 'pass 2 + SINR math, first link', 'pass 2 + SINR math, second link',
          'result stores issued (table rows: barrier + LDS round trip)', 'wave sum, ticket, (last wave) reward']
STAMPS = 8192                      # D2D_TUNE_STEP_ABLATE bit: stamps on


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--out', default='')
    ap.add_argument('--mode', default='none', choices=['none', 'table'])
    ap.add_argument('--lpt', type=int, default=2)
    args = ap.parse_args()
    b, c, p, r = 4096, 256, 256, 256
    if args.mode == 'none':
        env = VecD2DEnv({'num_rbs': r, 'num_cues': c, 'num_due_pairs': p, 'obs_fn': SignalPlanesObsFunction}, num_envs=b, reward_per_env=True)
    else:
        env = VecD2DEnv({'num_rbs': r, 'num_cues': c, 'num_due_pairs': p, 'obs_fn': OwnLinkObsFunction}, num_envs=b)
    env.reset(seed=1)
    h = env.simulator.handle
    h.set_export_actions(False)
    h.set_tuning(_native.TUNE_STEP_LPT, args.lpt)
    act = torch.randint(0, r * 21, (8, b, c + p), device=env.device, dtype=torch.int32)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

    def timed(ab):
        h.set_tuning(9, ab)                                   # D2D_TUNE_STEP_ABLATE (include/d2d_hip_diag.h)
        h.step(act[0].data_ptr())
        e0.record()
        for k in range(32):
            h.step(act[k % 8].data_ptr())
        e1.record(); e1.synchronize()
        return e0.elapsed_time(e1) / 32 * 1e3
    plain_us = float(np.median([timed(0) for _ in range(7)]))
    stamped_us = float(np.median([timed(STAMPS) for _ in range(7)]))
    h.step(act[7].data_ptr())
    torch.cuda.synchronize()
    waves = (c + p) // (64 * args.lpt)
    raw = np.zeros((b, 16, 16), dtype=np.uint64)
    lib = h._lib
    lib.d2d_debug_stamps.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    lib.d2d_debug_stamps.restype = C.c_int
    assert lib.d2d_debug_stamps(h._h, raw.ctypes.data_as(C.c_void_p), raw.nbytes) == 0
    st = raw.reshape(-1, 16)[: b * waves, :9].reshape(b, waves, 9).astype(np.int64)
    if args.lpt == 1:
        st[:, :, 6] = st[:, :, 5]                             # one link per thread: no second link
    # s_memtime counts on a base of its own per pair of CUs or so (offsets of up to milliseconds): cluster the workgroups by their
    # raw start, normalise per cluster, and keep for the TIMELINE only the clusters that are one clock domain (a span like the
    # launch's; two domains whose bases happen to lie close merge into a long one)
    first = st[:, :, 0].min(axis=1)
    order = np.argsort(first)
    dom = np.zeros(b, dtype=np.int64)
    dom[order] = np.cumsum(np.concatenate([[0], (np.diff(first[order]) > 100000).astype(np.int64)]))
    spans = np.zeros(int(dom.max()) + 1, dtype=np.int64)
    for x in range(spans.size):
        st[dom == x] -= st[dom == x][:, :, 0].min()
        spans[x] = st[dom == x][:, :, 8].max()
    good = np.isin(dom, np.nonzero(spans < 1.4 * np.median(spans))[0])
    span = int(spans[spans < 1.4 * np.median(spans)].max())
    d = np.diff(st, axis=2).reshape(-1, 8)
    life = (st[:, :, 8] - st[:, :, 0]).reshape(-1)
    out = {'kernel': 'rollout_kernel<0, 6, %d> 4096 x 512, obs mode %s, no decoded planes' % (args.lpt, args.mode),
           'clock': 's_memtime ticks, normalised per clock domain (%d single-domain clusters holding %d of the %d workgroups make the timeline)' % (int((spans < 1.4 * np.median(spans)).sum()), int(good.sum()), b),
           'launch_span_ticks': {'median': float(np.median(spans)), 'longest_single_domain': span},
           'launch_us_group_timed': round(plain_us, 2), 'launch_us_with_stamps': round(stamped_us, 2), 'ticks_per_us_if_span_is_the_stamped_launch': round(float(np.median(spans)) / stamped_us, 1),
           'wave_life_ticks': {'mean': float(life.mean()), 'median': float(np.median(life)), 'p90': float(np.percentile(life, 90))}, 'phases': []}
    for k, name in enumerate(PHASES):
        col = d[:, k]
        out['phases'].append({'phase': name, 'mean': round(float(col.mean()), 1), 'median': float(np.median(col)), 'p90': float(np.percentile(col, 90)),
                              'share_of_wave_life': round(float(col.mean() / life.mean()), 3)})
    wg_start = st[good][:, :, 0].min(axis=1); wg_end = st[good][:, :, 8].max(axis=1)
    out['workgroup_start_ticks_percentiles'] = {str(q): float(np.percentile(wg_start, q)) for q in (0, 10, 25, 50, 75, 90, 100)}
    out['workgroup_end_ticks_percentiles'] = {str(q): float(np.percentile(wg_end, q)) for q in (0, 10, 25, 50, 75, 90, 100)}
    out['workgroup_life_ticks'] = {'mean': float((wg_end - wg_start).mean()), 'median': float(np.median(wg_end - wg_start))}
    # how many workgroups of a domain start at once (residency) against how many it runs in all
    ok = np.nonzero(spans < 1.4 * np.median(spans))[0]
    out['per_domain_workgroups_total_and_started_in_the_first_2000_ticks'] = [[int((dom == x).sum()), int((st[dom == x][:, :, 0].min(axis=1) < 2000).sum())] for x in ok]
    flat = st[good].reshape(-1, 9)
    out['waves_in_phase_at_sampled_instants'] = [{'t': int(t), 'waves': [int(((flat[:, k] <= t) & (t < flat[:, k + 1])).sum()) for k in range(8)]}
                                                 for t in np.linspace(0, span, 25)[1:-1]]
    print(json.dumps(out, indent=1))
    if args.out:
        Path(args.out).write_text(json.dumps(out, indent=1))
    env.close()


if __name__ == '__main__':
    main()
