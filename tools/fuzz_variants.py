#!/usr/bin/env python3
"""Fuzz: random shapes x random launch / search / option settings of the step kernel must reproduce the all-pairs sweep of the
generic kernel BIT FOR BIT on every output (the committed tests pin chosen shapes; this walks the space for a few minutes).

    python tools/fuzz_variants.py [seconds]
"""
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / 'tests')]
from gym_d2d_amd import _native as nat
from gym_d2d_amd.simulator import Simulator
from sim_util import random_layout

OUTS = ('BUF_SINR_DB', 'BUF_SNR_DB', 'BUF_RATE_BPS', 'BUF_CAPACITY', 'BUF_REWARD', 'BUF_OBS_TABLE', 'BUF_RB', 'BUF_PWR', 'BUF_ENV_FLAGS')


def snap(sim, linear):
    out = {n: sim.fetch(getattr(nat, n)).copy() for n in OUTS}
    if linear:
        out['BUF_OBS'] = sim.fetch(nat.BUF_OBS).copy()
    return out


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    rng = np.random.default_rng(int(time.time()))
    t0, cases, runs = time.time(), 0, 0
    while time.time() - t0 < budget:
        kind = rng.integers(0, 4)
        if kind == 0:      # one link per thread, one env per workgroup (the rollout specialisation when N % 64 == 0)
            n = int(rng.choice([64, 128, 192, 256, 256, 512, 768, 1024])); cues = n // 2; dues = n - cues   # classes of 128 k links: two links per thread
        elif kind == 1:    # small envs sharing a workgroup
            cues, dues = int(rng.integers(0, 40)), int(rng.integers(1, 40))
        elif kind == 2:    # odd sizes
            hi = 300 if rng.random() < 0.6 else 520               # up to 1024 links: the padded rollout kernel; beyond: the generic lists
            cues, dues = int(rng.integers(1, hi)), int(rng.integers(1, hi))
        else:              # beyond the masks
            cues, dues = int(rng.integers(500, 1024)), int(rng.integers(525, 1024))
        rbs = int(rng.choice([1, 2, 5, 16, 64, 300, max(1, (cues + dues) // 2)]))
        b = int(rng.integers(1, 20)) if cues + dues > 600 else int(rng.integers(1, 70))
        reward = int(rng.integers(1, 4))
        linear = (cues + dues <= 128 and rng.random() < 0.5) or (cues + dues <= 400 and rng.random() < 0.25)   # beyond 128 links: the stand-alone expansion
        model = rng.choice(['log2', 'ple', 'ple', 'hata'])
        cfg = dict(num_rbs=rbs, num_cues=cues, num_due_pairs=dues, num_envs=b)
        if model == 'ple':
            from gym_d2d_amd.path_loss import LogDistancePathLoss
            # round 6: any exponent - PL_POWK for every k in 1 .. 8 (odd ones take a v_rsq on top), the general split beyond 8.5
            ple = float(rng.choice([1.2, 2.5, 3.0, 3.3, 3.5, 4.4, 5.6, 7.9, 9.1]))

            class Ple(LogDistancePathLoss):
                def __init__(self, f):
                    super().__init__(f, ple=ple)
            cfg['path_loss_model'] = Ple
        elif model == 'hata':
            from gym_d2d_amd.path_loss import CostHataPathLoss
            cfg['path_loss_model'] = CostHataPathLoss
        sim = Simulator(cfg)
        pos = random_layout(rng, b, cues, dues)
        if rng.random() < 0.3 and cues + dues > 8:
            # near / far geometry: some devices within 0.2 - 3 m of others, so that a receiver's same-RB terms span more than the
            # rollout kernel's exactness window (its sorted / selection / sweep fallbacks run; csrc/d2d_rollout.hip)
            d = pos.shape[1]
            for _ in range(int(rng.integers(1, 12))):
                e, i, j = int(rng.integers(0, b)), int(rng.integers(1, d)), int(rng.integers(1, d))
                if i != j:
                    pos[e, i] = pos[e, j] + (rng.uniform(0.2, 3.0) * np.array([np.cos(k := rng.uniform(0, 6.283)), np.sin(k)])).astype(np.float32)
        if rng.random() < 0.3:
            # round 6: a float64 layout off the float32 grid (d2d_set_positions_f64): the hi + lo variants of every kernel
            pos = pos.astype(np.float64)
            pos[:, 1:] += rng.uniform(-1e-5, 1e-5, pos[:, 1:].shape)
        sim.set_positions(pos)
        sim.set_links(sim.default_link_keys())
        p = sim.config.num_pwr_actions
        raw = np.concatenate([rng.integers(0, rbs * p['cue'], (b, cues)), rng.integers(0, rbs * p['due'], (b, dues))], 1).astype(np.int32)
        if rng.random() < 0.3 and cues + dues > 3:           # a skewed policy: most links on two RBs
            hot = rng.integers(0, cues + dues, (cues + dues) // 2)
            lv = np.array([p['cue']] * cues + [p['due']] * dues)
            raw[:, hot] = (rng.integers(0, min(2, rbs), (b, hot.size)) * lv[hot] + 3).astype(np.int32)
        if rng.random() < 0.2:
            raw[rng.integers(0, b), rng.integers(0, cues + dues)] = -7   # rb = -1: out of range
        h = sim.handle
        n_all = cues + dues
        fixed_idx = np.zeros(0, dtype=np.int64)
        if rng.random() < 0.35 and n_all > 2:              # traffic-model style links: (rb, pwr) held in the link records
            if rng.random() < 0.5:
                fixed_idx = np.arange(int(rng.integers(1, max(2, cues + 1))))                  # a prefix (the CUE block or part of it)
            else:
                fixed_idx = np.sort(rng.choice(n_all, size=int(rng.integers(1, n_all)), replace=False))   # an arbitrary set
            h.set_fixed_actions(fixed_idx, rng.integers(0, rbs, fixed_idx.size), rng.integers(0, 40, fixed_idx.size))
            keep = np.setdiff1d(np.arange(n_all), fixed_idx)
            raw = np.ascontiguousarray(raw[:, keep])
        h.set_obs_mode(nat.OBS_LINEAR if linear else nat.OBS_TABLE)
        h.set_reward(reward, {1: float(rng.choice([0.0, 0.4])), 2: -70.0, 3: 0.0}[reward])
        h.set_bucketing(False)
        sim.step_arrays(raw)
        ref = snap(sim, linear)
        h.set_bucketing(True)
        for _ in range(5):
            tune = {nat.TUNE_STEP_WALK: int(rng.choice([-1, 0, 2])), nat.TUNE_STEP_SCALAR_RECORDS: int(rng.choice([-1, 0, 1])),
                    nat.TUNE_STEP_NT_RESULTS: int(rng.choice([0, 1])), nat.TUNE_STEP_OBS_ROTATE: int(rng.choice([-1, 0, 7])),
                    nat.TUNE_STEP_LPT: int(rng.choice([-1, 1, 2])) if cues + dues <= 1024 else -1,
                    nat.TUNE_STEP_ENVS_PER_WG: int(rng.choice([0, 1, 2, 4])) if cues + dues <= 128 else 0,
                    nat.TUNE_STEP_FUSE_OBS: int(rng.choice([-1, 0, 1])) if linear and cues + dues <= 128 else -1,
                    # round 4: the stand-alone expansion's shapes (flat slabs / row-aligned / no LDS), slab sizes, store policies
                    nat.TUNE_OBS_BLOCK: int(rng.choice([0, 256, 512, 768, 1024])),
                    nat.TUNE_OBS_ROWS_PER_WG: int(rng.choice([0, 1, 2, 3, 4, 7])), nat.TUNE_OBS_NONTEMPORAL: int(rng.choice([1, 1, 0])),
                    nat.TUNE_STEP_BLOCK: int(rng.choice([0, 0, 256, 512, 1024])) if linear and cues + dues <= 64 else 0}
            if tune[nat.TUNE_STEP_LPT] == 1 and cues + dues > 1024:
                tune[nat.TUNE_STEP_LPT] = -1
            for k, v in tune.items():
                h.set_tuning(k, v)
            export = bool(rng.random() < 0.7)
            h.set_export_actions(export)
            if not export:
                h.upload(nat.BUF_RB, ref['BUF_RB']); h.upload(nat.BUF_PWR, ref['BUF_PWR'])
            # round 4: the per-env reward layout, the obs-less mode, float64 obs - each leaves every other output as it was
            per_env = reward == 1 and rng.random() < 0.4
            obs_none = (not linear) and rng.random() < 0.3
            obs64 = linear and rng.random() < 0.25
            h.set_reward_layout(nat.REWARD_PER_ENV if per_env else nat.REWARD_PER_AGENT)
            h.set_obs_mode(nat.OBS_NONE if obs_none else (nat.OBS_LINEAR if linear else nat.OBS_TABLE))
            h.set_obs_dtype(nat.F64 if obs64 else nat.F32)
            if per_env:
                h.upload(nat.BUF_REWARD, ref['BUF_REWARD'])
            if obs_none:
                h.upload(nat.BUF_OBS_TABLE, ref['BUF_OBS_TABLE'])
            sim.step_arrays(raw)
            got = snap(sim, linear)
            runs += 1
            if per_env and not np.array_equal(sim.fetch(nat.BUF_REWARD_ENV), ref['BUF_REWARD'][:, 0], equal_nan=True) and tune[nat.TUNE_STEP_LPT] != 2:
                print('MISMATCH per-env reward', dict(b=b, rbs=rbs, cues=cues, dues=dues), {int(k): v for k, v in tune.items()}, flush=True)
                sys.exit(1)
            if obs64:
                got['BUF_OBS'] = got['BUF_OBS'].astype(np.float32) if np.array_equal(got['BUF_OBS'], got['BUF_OBS'].astype(np.float32).astype(np.float64), equal_nan=True) else got['BUF_OBS'] * np.nan
            h.set_reward_layout(nat.REWARD_PER_AGENT); h.set_obs_dtype(nat.F32)
            h.set_obs_mode(nat.OBS_LINEAR if linear else nat.OBS_TABLE)
            for name, r in ref.items():
                if name == 'BUF_REWARD' and reward == 1 and tune[nat.TUNE_STEP_LPT] == 2 and cues + dues <= 1024:
                    # two links per thread add the capacities in another order than one link per thread: last-bit differences
                    ok = np.allclose(got[name], r, rtol=2e-6, atol=0.0, equal_nan=True)
                elif name == 'BUF_ENV_FLAGS':
                    ok = np.array_equal(got[name], r)
                else:
                    ok = np.array_equal(got[name], r, equal_nan=True)
                if not ok:
                    print('MISMATCH', name, dict(b=b, rbs=rbs, cues=cues, dues=dues, reward=reward, linear=linear, model=str(model), export=export),
                          {int(k): v for k, v in tune.items()}, flush=True)
                    sys.exit(1)
        h.set_export_actions(True)
        h.close()
        cases += 1
    # d2d_step_host on both sides of its zero-copy limit (256 KB of results: N = 100 links of one env with LinearObs is 245 KB,
    # N = 104 is 264 KB): the packed host block must equal d2d_step_rb_pwr + per-buffer downloads either way
    hosts = 0
    for n in list(range(94, 112)) + [30, 300]:
        for obs_mode in (nat.OBS_LINEAR, nat.OBS_TABLE):
            cues, dues, rbs, b = n // 2, n - n // 2, 9, (1 if obs_mode == nat.OBS_LINEAR else int(rng.integers(1, 60)))
            sim = Simulator(dict(num_rbs=rbs, num_cues=cues, num_due_pairs=dues, num_envs=b))
            sim.set_positions(random_layout(rng, b, cues, dues))
            sim.set_links(sim.default_link_keys())
            h = sim.handle
            h.set_obs_mode(obs_mode)
            rb = rng.integers(0, rbs, (b, n)).astype(np.int32); pw = rng.integers(0, 20, (b, n)).astype(np.int32)
            sim.step_arrays(rb=rb, pwr=pw)
            want = snap(sim, obs_mode == nat.OBS_LINEAR)
            res = h.step_host(rb, pw)
            for key, buf in (('sinr_db', 'BUF_SINR_DB'), ('snr_db', 'BUF_SNR_DB'), ('rate_bps', 'BUF_RATE_BPS'), ('capacity', 'BUF_CAPACITY'),
                             ('reward', 'BUF_REWARD'), ('obs_table', 'BUF_OBS_TABLE'), ('env_flags', 'BUF_ENV_FLAGS')) + \
                    ((('obs', 'BUF_OBS'),) if obs_mode == nat.OBS_LINEAR else ()):
                if not np.array_equal(res[key], want[buf]):
                    print('MISMATCH step_host', key, dict(n=n, b=b, obs_mode=obs_mode), flush=True)
                    sys.exit(1)
            assert (res['rb'] == rb).all() and (res['pwr'] == pw).all()
            h.close()
            hosts += 1
    print(f'step_host ok: {hosts} sizes around the zero-copy limit')
    print(f'fuzz ok: {cases} random cases, {runs} variant runs, all outputs bit-identical to the all-pairs sweep', flush=True)


if __name__ == '__main__':
    main()
