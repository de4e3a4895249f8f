#!/usr/bin/env python3
"""Fuzz of the drop-in single-env D2DEnv (dict in / dict out, d2d_env.py:62-116) against the oracle: random sizes, random episodes
in which every step a random SUBSET of links acts in a random order (uplinks, downlinks from the base station, sidelinks), actions
given as ints or as the reference's (2, 1) ndarray form, all three reward functions.  Checks the info dicts, the rewards and the
LinearObs rows (own link first, then the others in the step's agent order) of every step.

    python tools/fuzz_dropin.py [seconds]
"""
import random
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / 'tests')]
from gym_d2d_amd.envs import D2DEnv
from gym_d2d_amd.envs.reward_fn import CueSinrShannonRewardFunction, ShannonRewardFunction, SystemCapacityRewardFunction
from oracle import d2d_oracle as orc

TOL = 1e-5


def rel(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float(np.max(np.abs(a - b) / np.maximum(np.abs(b), 1.0))) if a.size else 0.0


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    rng = np.random.default_rng(int(time.time()))
    t0, envs, steps, worst = time.time(), 0, 0, 0.0
    while time.time() - t0 < budget:
        cues, dues, rbs = int(rng.integers(0, 12)), int(rng.integers(1, 12)), int(rng.integers(1, 9))
        rcls = [SystemCapacityRewardFunction, ShannonRewardFunction, CueSinrShannonRewardFunction][int(rng.integers(0, 3))]
        random.seed(int(rng.integers(0, 2 ** 31)))
        env = D2DEnv({'num_rbs': rbs, 'num_cues': cues, 'num_due_pairs': dues, 'reward_fn': rcls})
        env.reset()
        devs = env.simulator.devices
        ids, cfgs, is_bs = orc.device_configs(cues, dues)
        cols = orc.device_columns(cfgs, is_bs)
        pos = np.array([d.position.as_tuple() for d in devs.values()], dtype=np.float64)[None]
        index = {d: k for k, d in enumerate(devs.keys())}
        p = env.num_pwr_actions
        for _ in range(int(rng.integers(1, 12))):
            keys = []
            half = rbs // 2
            for c in range(cues):
                roll = rng.random()
                if roll < 0.5:
                    keys.append((f'cue{c:02d}', 'mbs'))
                elif roll < 0.7 and rbs > 1:
                    keys.append(('mbs', f'cue{c:02d}'))
            for k in range(dues):
                if rng.random() < 0.7:
                    keys.append((f'due{2 * k:02d}', f'due{2 * k + 1:02d}'))
            if not keys:
                keys = [('due00', 'due01')]
            order = rng.permutation(len(keys))
            keys = [keys[k] for k in order]
            raw, rb_l, pw_l, ty_l = {}, [], [], []
            for tx, rx in keys:
                kind = 'due' if tx.startswith('due') else ('cue' if tx.startswith('cue') else 'mbs')
                ty = {'due': 3, 'cue': 1, 'mbs': 2}[kind]
                # downlinks and uplinks never share an RB: both ends at the base station would be a zero distance
                lo, hi = (0, max(1, half)) if ty == 2 else ((half, rbs) if rbs > 1 else (0, 1))
                if ty == 3:
                    lo, hi = 0, rbs
                r, w = int(rng.integers(lo, hi)), int(rng.integers(0, p[kind]))
                rb_l.append(r); pw_l.append(w); ty_l.append(ty)
                raw[f'{tx}:{rx}'] = (r * p[kind] + w) if rng.random() < 0.7 else np.array([[r], [w]])
            obs, rewards, done, info = env.step(raw)
            tx_i = np.array([index[t] for t, _ in keys]); rx_i = np.array([index[r] for _, r in keys])
            rb = np.array([rb_l]); pw = np.array([pw_l]); ty = np.array(ty_l)
            ref = orc.step(pos, tx_i, rx_i, rb, pw, cols, orc.PathLossSpec())
            names = [f'{t}:{r}' for t, r in keys]
            assert list(obs) == names and list(rewards) == names and list(info) == names
            for f, g in (('sinr_db', 'sinr_db'), ('snr_db', 'snr_db'), ('rate_bps', 'rate_bps'), ('capacity_mbps', 'capacity_mbps')):
                e = rel([info[k][g] for k in names], ref[f][0]); worst = max(worst, e)
                assert e <= TOL, (f, e, keys)
            assert [info[k]['rb'] for k in names] == rb_l and [info[k]['tx_pwr_dbm'] for k in names] == pw_l
            if rcls is SystemCapacityRewardFunction:
                want = np.full(len(keys), orc.reward_system_capacity(ref['capacity_mbps'], rb, ty)[0])
            elif rcls is ShannonRewardFunction:
                want = orc.reward_shannon(ref['sinr_db'])[0]
            else:
                want = orc.reward_cue_sinr_shannon(ref['sinr_db'], rb, ty)[0]
            near = np.abs(ref['sinr_db'][0]) < 1e-3 if rcls is CueSinrShannonRewardFunction else np.zeros(len(keys), bool)
            if not near.any():
                assert rel([rewards[k] for k in names], want) <= TOL, (rcls.__name__, keys)
            table = orc.obs_table(pos, tx_i, rx_i, ref['sinr_db'], ref['snr_db'])
            assert rel(np.stack([obs[k] for k in names]), orc.expand_obs(table)[0]) <= TOL
            assert obs[names[0]].dtype == np.float64 and done == {'__all__': env.num_steps >= 10}
            steps += 1
        env.close()
        envs += 1
    print(f'drop-in fuzz ok: {envs} random envs, {steps} steps with random link subsets; worst info error {worst:.2e}', flush=True)


if __name__ == '__main__':
    main()
