#!/usr/bin/env python3
"""Generate golden vectors by RUNNING the reference (davidcotton/gym-d2d) in this container.

Usage (build container only - /root/reference does not exist on the GPU box):

    python tests/golden/make_golden.py            # rewrites tests/golden/*.npz

The reference needs `gym`, which is not installed; a throw-away stub (written to a temp dir
below, never shipped) supplies gym.Env / spaces / register / make.  gym contributes no
arithmetic to the path, so the stub cannot perturb results.

What is stored is DATA ONLY: inputs (device configs, float32-representable positions, raw
actions) and the reference's outputs (decoded rb/pwr, sinr/snr/rate/capacity from `info`,
rewards under all reward functions, obs rows).  No reference source travels.

Positions are rounded to float32 BEFORE the reference computes anything from them, so the
reference (fp64 arithmetic), the NumPy oracle and the fp32 HIP kernels all see identical
inputs - except in the `case16_unrounded_*` cases, which keep the float64 layouts exactly as
the reference's own reset() drew them (position.py:18-45): what a user of the reference has.
The HIP path takes those through d2d_set_positions_f64 (hi + lo float32 pairs).
"""
import json
import os
import random
import sys
import tempfile
import textwrap
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
REF_SRC = Path('/root/reference/src')

GYM_STUB = {
    'gym/__init__.py': '''
        from . import spaces
        from .spaces import Space
        from .envs.registration import register, make
        class Env:
            metadata = {}
            def reset(self): raise NotImplementedError
            def step(self, action): raise NotImplementedError
            def render(self, mode='human'): raise NotImplementedError
    ''',
    'gym/spaces.py': '''
        import numpy as np
        _rng = np.random.Generator(np.random.PCG64(0))
        def seed(s):
            global _rng
            _rng = np.random.Generator(np.random.PCG64(s))
        class Space:
            pass
        class Discrete(Space):
            def __init__(self, n): self.n = int(n)
            def sample(self): return int(_rng.integers(0, self.n))
            def contains(self, x): return 0 <= int(x) < self.n
        class Box(Space):
            def __init__(self, low, high, shape=None, dtype=np.float32):
                self.low, self.high, self.shape, self.dtype = low, high, tuple(shape), dtype
        class Dict(Space):
            def __init__(self, spaces): self.spaces = dict(spaces)
            def __getitem__(self, k): return self.spaces[k]
    ''',
    'gym/envs/__init__.py': '',
    'gym/envs/registration.py': '''
        import importlib
        _registry = {}
        def register(id, entry_point, **kw): _registry[id] = entry_point
        def make(id, **kwargs):
            mod, cls = _registry[id].split(':')
            return getattr(importlib.import_module(mod), cls)(**kwargs)
    ''',
}


def import_reference():
    tmp = tempfile.mkdtemp(prefix='gymstub_')
    for rel, body in GYM_STUB.items():
        p = Path(tmp) / rel
        p.parent.mkdir(parents=True, exist_ok=True)
        p.write_text(textwrap.dedent(body))
    sys.path.insert(0, str(REF_SRC))
    sys.path.insert(0, tmp)
    import gym                      # noqa: F401  (the stub)
    import gym_d2d                  # noqa: F401  registers D2DEnv-v0
    import gym_d2d.envs             # noqa: F401  (import order: envs before simulator)
    return gym


def f32(x):
    return float(np.float32(x))


def round_positions(env):
    """Overwrite every device position with its float32 rounding (see module docstring)."""
    from gym_d2d.position import Position
    for dev in env.simulator.devices.values():
        dev.set_position(Position(f32(dev.position.x), f32(dev.position.y)))


def recompute_after_reset(env):
    """Replay the tail of D2DEnv.reset (d2d_env.py:50-51) on the rounded positions."""
    env.actions._rbs.clear()
    env.state = env.simulator.step(env.actions)
    return env.obs_fn.get_state(env.actions, env.state, env.simulator.devices)


def snapshot_devices(env):
    ids, pos, cfgs, is_bs = [], [], [], []
    from gym_d2d.device import BaseStation
    for dev_id, dev in env.simulator.devices.items():
        ids.append(str(dev_id)); pos.append(dev.position.as_tuple()); cfgs.append(dict(dev.config))
        is_bs.append(isinstance(dev, BaseStation))
    return ids, np.asarray(pos, dtype=np.float64), cfgs, np.asarray(is_bs)


def record_step(env, extra_rewards, obs, rewards=None, game_over=None):
    """Collect everything observable about the step that just ran (env.actions / env.state)."""
    from gym_d2d.envs.reward_fn import ShannonRewardFunction, CueSinrShannonRewardFunction, \
        SystemCapacityRewardFunction
    acts = env.actions
    keys = [f'{t}:{r}' for (t, r) in acts.keys()]
    st = env.state
    rec = {
        'keys': keys,
        'rb': np.asarray([a.rb for a in acts.values()], dtype=np.int64),
        'pwr': np.asarray([a.tx_pwr_dBm for a in acts.values()], dtype=np.int64),
        'link_type': np.asarray([a.link_type.value for a in acts.values()], dtype=np.int64),
        'sinr_db': np.asarray([st['sinrs_db'][k] for k in acts.keys()]),
        'snr_db': np.asarray([st['snrs_db'][k] for k in acts.keys()]),
        'rate_bps': np.asarray([st['rate_bps'][k] for k in acts.keys()]),
        'capacity_mbps': np.asarray([st['capacity_mbps'][k] for k in acts.keys()]),
    }
    sysr = SystemCapacityRewardFunction()(acts, st)
    rec['reward_system_capacity'] = np.asarray([sysr[k] for k in keys])
    sh = ShannonRewardFunction()(acts, st)
    rec['reward_shannon'] = np.asarray([sh[k] for k in keys])
    cs = CueSinrShannonRewardFunction()(acts, st)
    rec['reward_cue_sinr_shannon'] = np.asarray([cs[k] for k in keys])
    for name, fn in extra_rewards.items():
        r = fn(acts, st)
        rec[f'reward_{name}'] = np.asarray([r[k] for k in keys])
    if rewards is not None:
        rec['reward_env'] = np.asarray([rewards[k] for k in keys])
    if game_over is not None:
        rec['game_over'] = np.asarray(bool(game_over['__all__']))
    n = len(keys)
    if obs is not None and n:
        assert list(obs.keys()) == keys
        rows = sorted({0, n // 2, n - 1})
        rec['obs_rows'] = np.asarray(rows, dtype=np.int64)
        rec['obs'] = np.stack([obs[keys[i]] for i in rows])
        rec['obs_table'] = np.stack([obs[k][:6] for k in keys])
        assert rec['obs'].dtype == np.float64
    return rec


def save_case(name, meta, devices, steps):
    # `python make_golden.py case14 ...` rewrites only the named cases: the reference sums interferers in set (hash)
    # order, so regenerating a file changes its last bits (~1e-16) and would churn the repository for nothing
    if len(sys.argv) > 1 and not any(tag in name for tag in sys.argv[1:]):
        return
    ids, pos, cfgs, is_bs = devices
    arrays = {'dev_pos': pos, 'dev_is_bs': is_bs}
    meta = dict(meta, dev_ids=ids, dev_cfgs=cfgs, num_steps=len(steps))
    for k, rec in enumerate(steps):
        meta[f's{k}_keys'] = rec.pop('keys')
        for field, arr in rec.items():
            arrays[f's{k}_{field}'] = np.asarray(arr)
    arrays['meta_json'] = np.frombuffer(json.dumps(meta, default=str).encode(), dtype=np.uint8)
    out = HERE / f'{name}.npz'
    np.savez_compressed(out, **arrays)
    print(f'wrote {out.name}: {len(steps)} steps, {len(ids)} devices, {out.stat().st_size} bytes')


class _UniformFeeder:
    """Stands in for the `random` module inside gym_d2d.position: random() hands out u[d, t, 0] (theta) then u[d, t, 1]
    (radius) for the device / try the sampler is at."""

    def __init__(self):
        self.u = None; self.d = 0; self.t = 0; self.k = 0; self.max_t = 0

    def start_device(self):
        self.d += 1; self.t = 0; self.k = 0

    def random(self):
        v = float(self.u[self.d, self.t, self.k])
        self.k += 1
        if self.k == 2:
            self.k = 0; self.t += 1; self.max_t = max(self.max_t, self.t)
        return v


def save_sampler_case(gym):
    if len(sys.argv) > 1 and not any(tag in 'sampler_case15' for tag in sys.argv[1:]):
        return
    sys.path.insert(0, str(HERE.parent.parent))
    from oracle import d2d_oracle as orc
    import gym_d2d.position as ref_position
    import gym_d2d.simulator as ref_simulator
    feeder = _UniformFeeder()
    orig = (ref_position.random, ref_simulator.get_random_position, ref_simulator.get_random_position_nearby)

    def wrap(fn):
        def inner(*a, **kw):
            feeder.start_device()
            return fn(*a, **kw)
        return inner
    out = {}
    meta = {'case': 'sampler_case15', 'configs': []}
    try:
        ref_position.random = feeder
        ref_simulator.get_random_position = wrap(orig[1])
        ref_simulator.get_random_position_nearby = wrap(orig[2])
        for tag, cues, dues, radius, d2d, seed, episode, first_env, envs in (
                ('default_radii', 7, 9, 500.0, 20.0, 0x1234_5678_9ABC, 3, 1000, 48),
                ('tight_cell', 3, 12, 30.0, 25.0, 77, 0, 0, 48)):              # d2d radius ~ cell radius: many rejections
            d = 1 + cues + 2 * dues
            u = orc.reset_uniforms(seed, episode, envs, d, 64, first_env=first_env)
            pos = np.zeros((envs, d, 2))
            env = gym.make('D2DEnv-v0', env_config={'num_cues': cues, 'num_due_pairs': dues, 'cell_radius_m': radius,
                                                    'd2d_radius_m': d2d})
            for e in range(envs):
                feeder.u = u[e]; feeder.d = 0; feeder.t = 0; feeder.k = 0
                env.simulator.reset()
                assert feeder.d == d - 1, (feeder.d, d)
                pos[e] = [dev.position.as_tuple() for dev in env.simulator.devices.values()]
            tries = feeder.max_t
            out[f'{tag}_u'] = u[:, :, :max(tries, 1) + 1]
            out[f'{tag}_pos'] = pos
            meta['configs'].append(dict(tag=tag, num_cues=cues, num_due_pairs=dues, cell_radius_m=radius, d2d_radius_m=d2d,
                                        seed=seed, episode=episode, first_env=first_env, num_envs=envs, max_tries_used=tries))
    finally:
        ref_position.random, ref_simulator.get_random_position, ref_simulator.get_random_position_nearby = orig
    out['meta_json'] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    path = HERE / 'sampler_case15.npz'
    np.savez_compressed(path, **out)
    print(f'wrote {path.name}: {[c["tag"] for c in meta["configs"]]}, max tries {[c["max_tries_used"] for c in meta["configs"]]}, '
          f'{path.stat().st_size} bytes')


def seed_all(gym, k):
    random.seed(k)
    gym.spaces.seed(k)


def sample_actions(env, keys):
    raw = {}
    for key in keys:
        tx = key.split(':')[0]
        kind = 'due' if tx.startswith('due') else ('cue' if tx.startswith('cue') else 'mbs')
        raw[key] = env.action_space[kind].sample()
    return raw


def env_meta(env, pl):
    c = env.simulator.config
    return {
        'num_rbs': c.num_rbs, 'num_cues': c.num_cues, 'num_due_pairs': c.num_due_pairs,
        'cell_radius_m': c.cell_radius_m, 'd2d_radius_m': c.d2d_radius_m,
        'due_min_tx_power_dBm': c.due_min_tx_power_dBm, 'due_max_tx_power_dBm': c.due_max_tx_power_dBm,
        'cue_max_tx_power_dBm': c.cue_max_tx_power_dBm, 'mbs_max_tx_power_dBm': c.mbs_max_tx_power_dBm,
        'carrier_freq_GHz': c.carrier_freq_GHz, 'num_subcarriers': c.num_subcarriers,
        'subcarrier_spacing_kHz': c.subcarrier_spacing_kHz, 'path_loss': pl,
        'num_pwr_actions': env.num_pwr_actions,
    }


def run_episode(gym, name, seed, env_config, pl, n_steps, *, extra_rewards=None, action_fn=None, raw_store=True, round_pos=True):
    """reset() + n_steps step()s with all links acting; stores raw int actions per step."""
    seed_all(gym, seed)
    env = gym.make('D2DEnv-v0', env_config=dict(env_config))
    env.reset()
    if round_pos:
        round_positions(env)
    obs = recompute_after_reset(env)
    extra_rewards = extra_rewards or {}
    steps = [record_step(env, extra_rewards, obs)]
    steps[0]['raw'] = np.asarray([-1] * len(obs), dtype=np.int64)      # reset actions exist only as (rb, pwr)
    for s in range(n_steps):
        keys = list(obs.keys())
        raw = action_fn(env, keys, s) if action_fn else sample_actions(env, keys)
        obs, rewards, game_over, info = env.step(raw)
        rec = record_step(env, extra_rewards, obs, rewards, game_over)
        # info mirrors state (d2d_env.py:106-116)
        assert [info[k]['sinr_db'] for k in rec['keys']] == list(rec['sinr_db'])
        first = next(iter(raw.values()))
        if isinstance(first, np.ndarray):
            rec['raw_rb_pwr'] = np.stack([np.asarray(raw[k]).reshape(2) for k in rec['keys']]).astype(np.int64)
        else:
            rec['raw'] = np.asarray([raw[k] for k in rec['keys']], dtype=np.int64)
        steps.append(rec)
    save_case(name, dict(env_meta(env, pl), seed=seed, case=name), snapshot_devices(env), steps)
    return env


def main():
    gym = import_reference()
    from gym_d2d.path_loss import LogDistancePathLoss, CostHataPathLoss, AreaType, PathLoss
    from gym_d2d.envs.reward_fn import SystemCapacityRewardFunction
    ld = {'kind': 'log_distance', 'ple': 2.0}

    # (1) default 25/25/25: reset + 10 steps (game_over flips on step 10, d2d_env.py:68)
    run_episode(gym, 'case01_default', 101, {}, ld, 10)

    # (2) 4 RB / 10 CUE / 30 DUE: heavy RB collisions
    run_episode(gym, 'case02_collisions', 102, {'num_rbs': 4, 'num_cues': 10, 'num_due_pairs': 30}, ld, 3)

    # (3) stress 256/256/256, 3 steps
    run_episode(gym, 'case03_stress256', 103, {'num_rbs': 256, 'num_cues': 256, 'num_due_pairs': 256}, ld, 3)

    # (4) every link on ONE resource block (worst-case O(N^2) interference)
    def one_rb(env, keys, s):
        out = {}
        for k in keys:
            kind = 'due' if k.startswith('due') else 'cue'
            p = env.num_pwr_actions[kind]
            out[k] = 7 * p + (env.action_space[kind].sample() % p)
        return out
    run_episode(gym, 'case04_one_rb', 104, {}, ld, 2, action_fn=one_rb)

    # (5) DUE-only subset of links (obs length shrinks to 6*len(actions), obs_fn.py:43-53)
    def due_only(env, keys, s):
        due = [f'{t}:{r}' for (t, r) in env.simulator.devices.dues.keys()]
        return sample_actions(env, due[::2] if s else due)
    run_episode(gym, 'case05_due_subset', 105, {'num_rbs': 6}, ld, 2, action_fn=due_only)

    # (6) DOWNLINK 'mbs:cueXX' links mixed with uplink + sidelink (d2d_env.py:88-90)
    # NB an uplink and a downlink on the SAME rb make the BS interfere with itself at distance 0, which the
    # reference turns into `ValueError: math domain error` (path_loss.py:66); so step 0 has no uplinks and
    # step 1 keeps uplinks (rb 2-4) and downlinks (rb 0-1) on disjoint resource blocks.
    def with_downlink(env, keys, s):
        cues = list(env.simulator.devices.cues.keys())
        dues = [f'{t}:{r}' for (t, r) in env.simulator.devices.dues.keys()]
        if s == 0:
            return sample_actions(env, [f'mbs:{c}' for c in cues[:6]] + dues[:10])
        out = {}
        pm, pc = env.num_pwr_actions['mbs'], env.num_pwr_actions['cue']
        for k, c in enumerate(cues[:5]):
            out[f'mbs:{c}'] = (k % 2) * pm + env.action_space['mbs'].sample() % pm
        for k, c in enumerate(cues[5:12]):
            out[f'{c}:mbs'] = (2 + k % 3) * pc + env.action_space['cue'].sample() % pc
        out.update(sample_actions(env, dues[:10]))
        return out
    run_episode(gym, 'case06_downlink', 106, {'num_rbs': 5}, ld, 2, action_fn=with_downlink)

    # (7) device_config_file: fixed positions + per-device overrides, save -> load round trip
    seed_all(gym, 107)
    env = gym.make('D2DEnv-v0', env_config={'num_rbs': 8, 'num_cues': 6, 'num_due_pairs': 6})
    env.reset()
    round_positions(env)
    tmp = Path(tempfile.mkdtemp()) / 'device_config.json'
    env.save_device_config(tmp)
    cfg_json = json.loads(tmp.read_text())
    for dev_id in ('cue03', 'due04', 'due05', 'mbs'):
        cfg_json[dev_id]['config'] = dict(cfg_json[dev_id]['config'])
    cfg_json['cue03']['config'].update(tx_antenna_gain_dBi=2.5, body_loss_dB=1.0, thermal_noise_dBm=-101.0)
    cfg_json['due04']['config'].update(ix_margin_dB=1.5, subcarrier_spacing_kHz=30)
    cfg_json['due05']['config'].update(rx_antenna_gain_dBi=3.0, noise_figure_dB=5.0, sinr_dB=-6.0)
    cfg_json['mbs']['config'].update(rx_antenna_gain_dBi=15.0, cable_loss_dB=3.0)
    del cfg_json['cue01'], cfg_json['due02'], cfg_json['due03']      # these stay random at reset
    tmp.write_text(json.dumps(cfg_json))
    run_episode(gym, 'case07_device_config', 1107,
                {'num_rbs': 8, 'num_cues': 6, 'num_due_pairs': 6, 'device_config_file': tmp}, ld, 2)
    (HERE / 'case07_device_config.json').write_text(json.dumps(cfg_json))

    # (8) COST-Hata, urban and suburban (path_loss.py:90-123)
    class UrbanHata(CostHataPathLoss):
        def __init__(self, f):
            super().__init__(f, AreaType.URBAN)
    run_episode(gym, 'case08_hata_urban', 108, {'num_rbs': 10, 'path_loss_model': UrbanHata},
                {'kind': 'cost_hata', 'area': 'urban'}, 2)
    run_episode(gym, 'case08_hata_suburban', 1108, {'num_rbs': 10, 'path_loss_model': CostHataPathLoss},
                {'kind': 'cost_hata', 'area': 'suburban'}, 2)

    # (9) LogDistance subclass, ple = 3.5
    class Ple35(LogDistancePathLoss):
        def __init__(self, f):
            super().__init__(f, ple=3.5)
    run_episode(gym, 'case09_ple35', 109, {'num_rbs': 10, 'path_loss_model': Ple35},
                {'kind': 'log_distance', 'ple': 3.5}, 2)

    # (10) user-defined PathLoss in the style of examples/custom_path_loss.py (Python-plugin route)
    from math import log10

    class FooPathLoss(PathLoss):
        def __call__(self, tx, rx):
            d = tx.position.distance(rx.position)
            return 20 * log10(d) - tx.tx_antenna_gain_dBi - rx.rx_antenna_gain_dBi
    env = run_episode(gym, 'case10_custom_pl', 110, {'num_rbs': 10, 'path_loss_model': FooPathLoss},
                      {'kind': 'custom_foo'}, 2)

    # (11) SystemCapacity with min_capacity_mbps > 0 so the -1 branch fires (reward_fn.py:38-39)
    extra = {f'system_capacity_min{str(m).replace(".", "p")}': SystemCapacityRewardFunction(min_capacity_mbps=m)
             for m in (0.05, 0.5, 2.0)}
    run_episode(gym, 'case11_min_capacity', 111, {'num_rbs': 8}, ld, 4, extra_rewards=extra)

    # (12) ndarray (2,1) action form: rb, pwr given explicitly (d2d_env.py:97-98)
    def array_actions(env, keys, s):
        out = {}
        for k in keys:
            kind = 'due' if k.startswith('due') else 'cue'
            a = env.action_space[kind].sample()
            p = env.num_pwr_actions[kind]
            out[k] = np.array([[a // p], [a % p]])
        return out
    run_episode(gym, 'case12_array_actions', 112, {}, ld, 2, action_fn=array_actions)

    # (13) ShadowingPathLoss: stochastic (a fresh gauss per call, path_loss.py:79), so what is captured is the
    # per-link DISTRIBUTION of the reference's outputs over many steps with the same positions and actions.
    from gym_d2d.path_loss import ShadowingPathLoss
    seed_all(gym, 113)
    env = gym.make('D2DEnv-v0', env_config={'num_rbs': 6, 'num_cues': 8, 'num_due_pairs': 12,
                                            'path_loss_model': ShadowingPathLoss})
    obs = env.reset()
    round_positions(env)
    raw = sample_actions(env, list(obs.keys()))
    reps = 2000
    sinr = np.empty((reps, len(raw))); snr = np.empty_like(sinr)
    for k in range(reps):
        _, _, _, info = env.step(raw)
        sinr[k] = [info[key]['sinr_db'] for key in raw]
        snr[k] = [info[key]['snr_db'] for key in raw]
    rec = record_step(env, {}, None)
    for f in ('sinr_db', 'snr_db', 'rate_bps', 'capacity_mbps', 'reward_system_capacity', 'reward_shannon',
              'reward_cue_sinr_shannon'):
        rec.pop(f)          # single draws are meaningless as pins
    rec.update(raw=np.asarray([raw[k] for k in rec['keys']], dtype=np.int64), reps=np.asarray(reps),
               sinr_mean=sinr.mean(0), sinr_std=sinr.std(0), snr_mean=snr.mean(0), snr_std=snr.std(0),
               diff_std=(sinr - snr).std(0), sinr_snr_corr=np.array([np.corrcoef(sinr[:, i], snr[:, i])[0, 1]
                                                                     for i in range(sinr.shape[1])]))
    save_case('case13_shadowing', dict(env_meta(env, {'kind': 'shadowing', 'ple': 2.0, 'd0_m': 100.0, 'chi_dB': 2.7}),
                                       seed=113, case='case13_shadowing'), snapshot_devices(env), [rec])

    # (14) CUE links driven by the reference's UplinkTrafficModel (traffic_model.py:15-22; constructed at
    # simulator.py:58 but never called by the reference's step) + agent-supplied DUE actions: SURVEY 8(f) rank 2.
    from gym_d2d.actions import Actions
    seed_all(gym, 114)
    env = gym.make('D2DEnv-v0', env_config={'num_rbs': 7, 'num_cues': 12, 'num_due_pairs': 16})
    env.reset()
    round_positions(env)
    steps = []
    for k in range(3):
        traffic = env.simulator.traffic_model.get_traffic(env.simulator.devices)
        due_raw = sample_actions(env, [f'{t}:{r}' for (t, r) in env.simulator.devices.dues.keys()])
        merged = Actions({**traffic.data, **env._extract_actions(due_raw).data})
        env.actions = merged
        env.state = env.simulator.step(merged)
        obs = env.obs_fn.get_state(merged, env.state, env.simulator.devices)
        rec = record_step(env, {}, obs)
        rec['due_raw'] = np.asarray(list(due_raw.values()), dtype=np.int64)
        steps.append(rec)
    save_case('case14_traffic_model', dict(env_meta(env, ld), seed=114, case='case14_traffic_model'),
              snapshot_devices(env), steps)

    # (16) the reference's OWN float64 layouts, not rounded: cases 01, 03 and 07 again with the positions exactly as reset() drew
    # them (position.py:18-45) resp. as save_device_config wrote them (d2d_env.py:124-134)
    run_episode(gym, 'case16_unrounded_default', 101, {}, ld, 10, round_pos=False)
    run_episode(gym, 'case16_unrounded_stress256', 103, {'num_rbs': 256, 'num_cues': 256, 'num_due_pairs': 256}, ld, 3, round_pos=False)
    seed_all(gym, 107)
    env = gym.make('D2DEnv-v0', env_config={'num_rbs': 8, 'num_cues': 6, 'num_due_pairs': 6})
    env.reset()
    tmp = Path(tempfile.mkdtemp()) / 'device_config.json'
    env.save_device_config(tmp)
    cfg_json = json.loads(tmp.read_text())
    for dev_id in ('cue03', 'due04', 'due05', 'mbs'):
        cfg_json[dev_id]['config'] = dict(cfg_json[dev_id]['config'])
    cfg_json['cue03']['config'].update(tx_antenna_gain_dBi=2.5, body_loss_dB=1.0, thermal_noise_dBm=-101.0)
    cfg_json['due04']['config'].update(ix_margin_dB=1.5, subcarrier_spacing_kHz=30)
    cfg_json['due05']['config'].update(rx_antenna_gain_dBi=3.0, noise_figure_dB=5.0, sinr_dB=-6.0)
    cfg_json['mbs']['config'].update(rx_antenna_gain_dBi=15.0, cable_loss_dB=3.0)
    del cfg_json['cue01'], cfg_json['due02'], cfg_json['due03']      # these stay random at reset
    tmp.write_text(json.dumps(cfg_json))
    run_episode(gym, 'case16_unrounded_device_config', 1107,
                {'num_rbs': 8, 'num_cues': 6, 'num_due_pairs': 6, 'device_config_file': tmp}, ld, 2, round_pos=False)
    if len(sys.argv) == 1 or any(tag in 'case16_unrounded_device_config' for tag in sys.argv[1:]):
        (HERE / 'case16_unrounded_device_config.json').write_text(json.dumps(cfg_json))

    # (15) the reference's own samplers - get_random_position / get_random_position_nearby (position.py:18-45) as driven
    # by Simulator.reset (simulator.py:61-75) - fed with the counter-based Philox uniforms the device-side reset
    # consumes (oracle.reset_uniforms) instead of Python's global Mersenne Twister: pins the oracle's sampler (and through
    # it csrc/d2d_reset.hip) to the reference's arithmetic and call order, not just to its distributional properties.
    save_sampler_case(gym)

    # known-answer values copied as DATA from the reference's own unit tests (file:line in the key)
    kat = {
        'test_path_loss.py:11 pl_constant_dB(2.1,2.0)': 38.892169116561746,
        'test_path_loss.py:25 logdist 2.1GHz 250m': 86.85097,
        'test_path_loss.py:27 logdist 2.1GHz 500m': 92.87156,
        'test_path_loss.py:48 hata urban bs->ue 250m': 121.44557455875727,
        'test_path_loss.py:49 hata urban ue->bs 250m': 114.35415557446962,
        'test_path_loss.py:51 hata urban bs->ue 500m': 132.2768393081241,
        'test_path_loss.py:52 hata urban ue->bs 500m': 127.5231950610599,
        'test_conversion.py:8 dB_to_linear(1)': 1.258925,
        'test_conversion.py:17 linear_to_dB(2)': 3.0103,
        'test_conversion.py:31 dBm_to_W(30)': 1.0,
        'test_conversion.py:37 W_to_dBm(0.2)': 23.0103,
        'test_device.py:73-77 ue eirp(12)': 12 + 0 - 3 - 3,
        'test_device.py:80-85 bs eirp(46)': 46 + 17.5 - 2 - 2 + 2,
    }
    # cross-check those against the imported reference before trusting them as pins
    from gym_d2d.path_loss import pl_constant_dB
    assert abs(pl_constant_dB(2.1, 2.0) - kat['test_path_loss.py:11 pl_constant_dB(2.1,2.0)']) < 1e-12
    (HERE / 'known_answers.json').write_text(json.dumps(kat, indent=1))
    print('done')


if __name__ == '__main__':
    main()
