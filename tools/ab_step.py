#!/usr/bin/env python3
"""A/B sweeps of the step kernel, interleaved rounds in ONE process (HIP events around every launch):

    python tools/ab_step.py stress     # 4096 x 512, compact-obs mode: mask walk vs all-pairs, per reward fn (the RB-sorted
                                       # variant of round 2 lives in git history; its A/B is profiles/r2_ab_step_variants_stress.jsonl)
    python tools/ab_step.py default    # 1024 x 50, LinearObs: envs per workgroup x fused obs x block size
    python tools/ab_step.py halves     # 1024 x 50: one env of 1024 vs two of 512 on one / two streams (wall clock per 1024 env-steps)

Prints one JSON line per variant; `--out file` also appends them to a file (profiles/r2_ab_*.jsonl are these).
"""
import argparse
import json
import statistics
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch

from gym_d2d_amd import _native
from gym_d2d_amd.envs import VecD2DEnv
from gym_d2d_amd.envs.obs_fn import LinearObsFunction, OwnLinkObsFunction


def emit(rec, out):
    line = json.dumps(rec)
    print(line, flush=True)
    if out:
        with open(out, 'a') as f:
            f.write(line + '\n')


def timed(h, act, launches, kernel=0):
    """Average GPU time per launch in us: ONE event pair around `launches` back-to-back launches (an event pair around a
    single launch adds 3-6 us of its own, tools/probes/launch_floor.hip); inter-launch gaps are included.
    act: one [B, A] action tensor (re-read every launch: it stays in the Infinity Cache) or a [K, B, A] stack that is
    walked round-robin - fresh actions every launch, as in a rollout (what bench.py does).
    kernel = 1 keeps the library's per-launch events (two kernels per step: the obs kernel's own time)."""
    if kernel == 1:
        h.profile_reset(); h.profile_enable(True)
        for k in range(launches):
            h.step((act[k % act.shape[0]] if act.dim() == 3 else act).data_ptr())
        ms, k = h.profile_read(kernel)
        h.profile_enable(False)
        return ms / max(k, 1) * 1e3
    stream = torch.cuda.current_stream()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    h.step((act[0] if act.dim() == 3 else act).data_ptr())
    e0.record(stream)
    for k in range(launches):
        h.step((act[k % act.shape[0]] if act.dim() == 3 else act).data_ptr())
    e1.record(stream)
    e1.synchronize()
    return e0.elapsed_time(e1) / launches * 1e3


def stress(args):
    b, c, p, r = 4096, 256, 256, 256
    env = VecD2DEnv({'num_rbs': r, 'num_cues': c, 'num_due_pairs': p, 'obs_fn': OwnLinkObsFunction}, num_envs=b)
    env.reset(seed=1)
    h = env.simulator.handle
    act = torch.randint(0, r * 21, (64, b, c + p), device=env.device, dtype=torch.int32)      # fresh actions per launch
    # (reward fn, variant, decoded rb / pwr exported).  Rollout-kernel options: scalar record loads (srec), nontemporal result
    # stores (nt); search variants: mask walk (default), member lists (generic kernel), all-pairs sweep.  'two_per_rb' = actions
    # that put exactly two links on every RB (no imbalance between the lanes of a wave).
    variants = [(1, name, ex) for name in ('plain', 'srec', 'nt', 'srec_nt') for ex in (1, 0)]
    if not args.quick:
        variants += [(rw, name, 1) for rw in (0, 2, 3) for name in ('srec_nt', 'member_lists')]
        variants += [(1, 'member_lists', 1), (1, 'all_pairs', 1), (1, 'srec_nt_two_per_rb', 1)]
    pc, pd = env.num_pwr_actions['cue'], env.num_pwr_actions['due']
    two = torch.cat([torch.arange(c, device=env.device, dtype=torch.int32) * pc + 3,
                     torch.arange(p, device=env.device, dtype=torch.int32) * pd + 5])[None, None].expand(64, b, c + p).contiguous()
    times = {v: [] for v in variants}
    for rnd in range(args.rounds):
        for v in variants:
            if v[1] == 'all_pairs' and rnd > 1:
                continue
            h.set_reward(v[0], {0: 0.0, 1: 0.0, 2: -70.0, 3: 0.0}[v[0]])
            h.set_bucketing(v[1] != 'all_pairs')
            h.set_export_actions(bool(v[2]))
            h.set_tuning(_native.TUNE_STEP_WALK, 2 if v[1].startswith('member_lists') else 0)
            h.set_tuning(_native.TUNE_STEP_SCALAR_RECORDS, int('srec' in v[1]))
            h.set_tuning(_native.TUNE_STEP_NT_RESULTS, int('nt' in v[1]))
            times[v].append(timed(h, two if v[1].endswith('two_per_rb') else act, 32 if v[1] != 'all_pairs' else 8))
    h.set_export_actions(True)
    bytes_per = b * (c + p) * 64.0
    for v in variants:
        med = statistics.median(times[v])
        emit({'sweep': 'stress_table_mode', 'reward_fn': v[0], 'variant': v[1], 'export_rb_pwr': v[2], 'median_us': round(med, 2),
              'min_us': round(min(times[v]), 2), 'algorithmic_GBps': round(bytes_per / med / 1e3), 'rounds': len(times[v])},
             args.out)
    env.close()


def default(args):
    b, c, p, r = 1024, 25, 25, 25
    env = VecD2DEnv({'num_rbs': r, 'num_cues': c, 'num_due_pairs': p, 'obs_fn': LinearObsFunction}, num_envs=b,
                    cue_actions='traffic')
    env.reset(seed=1)
    h = env.simulator.handle
    act = env.action_buffer()
    variants = [(epw, fuse, blk, var) for var in (0,) for fuse in (1, 0) for epw in (1, 2, 4, 8)
                for blk in ((0, 256, 512, 1024) if fuse else (0,)) if blk == 0 or blk >= epw * 64]
    times = {v: [] for v in variants}
    for rnd in range(args.rounds):
        for v in variants:
            h.set_tuning(_native.TUNE_STEP_ENVS_PER_WG, v[0])
            h.set_tuning(_native.TUNE_STEP_FUSE_OBS, v[1])
            h.set_tuning(_native.TUNE_STEP_BLOCK, v[2])
            t_all = timed(h, act, 20, 0)                      # whole step (one or two kernels), group timed
            t_obs = 0.0 if v[1] else timed(h, act, 20, 1)    # the obs kernel alone (per-launch events), two-launch form only
            times[v].append((t_all - t_obs, t_obs))
    for v in variants:
        s_med = statistics.median(t[0] for t in times[v]); o_med = statistics.median(t[1] for t in times[v])
        emit({'sweep': 'default_linear_obs', 'envs_per_wg': v[0], 'fuse_obs': v[1], 'block': v[2], 'variant': v[3],
              'step_kernel_us': round(s_med, 2), 'obs_kernel_us': round(o_med, 2), 'sum_us': round(s_med + o_med, 2)}, args.out)
    env.close()


def halves(args):
    """BASELINE config 2 (1024 x 50, fused LinearObs) as ONE env of 1024 against TWO envs of 512 - on one stream, and on two streams
    so that one half's 30 MB of stores can run under the other half's latency-bound step phase (the fused kernel
    is launch 1.5 us + step phase 3.0 us + 8.9 us of stores, every workgroup in the same phase at the same time).  Wall time per
    1024 env-steps, fresh actions, interleaved rounds, 2000 steps per sample."""
    cfg = {'num_rbs': 25, 'num_cues': 25, 'num_due_pairs': 25, 'obs_fn': LinearObsFunction}
    one = VecD2DEnv(dict(cfg), num_envs=1024, cue_actions='traffic'); one.reset(seed=1)
    streams = [torch.cuda.Stream() for _ in range(2)]
    same = [VecD2DEnv(dict(cfg), num_envs=512, cue_actions='traffic', first_env=512 * k) for k in range(2)]
    for e in same:
        e.reset(seed=1)
    split = []
    for k in range(2):
        with torch.cuda.stream(streams[k]):
            e = VecD2DEnv(dict(cfg), num_envs=512, cue_actions='traffic', first_env=512 * k); e.reset(seed=1); split.append(e)
    torch.cuda.synchronize()
    acts = torch.randint(0, 25 * 21, (64, 1024, 25), device=one.device, dtype=torch.int32)

    def run(name, n):
        for k in range(n):
            a = acts[k % 64]
            if name == 'one env of 1024':
                one.step(a)
            elif name == 'two envs of 512, one stream':
                same[0].step(a[:512]); same[1].step(a[512:])
            else:
                for j in range(2):
                    with torch.cuda.stream(streams[j]):
                        split[j].step(a[512 * j:512 * (j + 1)])
    names = ('one env of 1024', 'two envs of 512, one stream', 'two envs of 512, two streams')
    times = {n: [] for n in names}
    for rnd in range(args.rounds):
        for n in names:
            run(n, 200); torch.cuda.synchronize()
            t0 = time.perf_counter()
            run(n, 2000); torch.cuda.synchronize()
            times[n].append((time.perf_counter() - t0) / 2000 * 1e6)
    for n in names:
        emit({'sweep': 'default_half_batches', 'variant': n, 'wall_us_per_1024_env_steps': round(statistics.median(times[n]), 2),
              'min': round(min(times[n]), 2)}, args.out)


if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('what', choices=['stress', 'default', 'halves'])
    ap.add_argument('--rounds', type=int, default=7)
    ap.add_argument('--quick', action='store_true', help='stress: only the mask walk and the member lists')
    ap.add_argument('--out', default='')
    a = ap.parse_args()
    {'stress': stress, 'default': default, 'halves': halves}[a.what](a)
