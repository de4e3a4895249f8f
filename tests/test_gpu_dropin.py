"""Drop-in behaviour of D2DEnv beyond numbers: plugins written against the reference's ABCs, the device-config
JSON round trip, render(), the example scripts.  `pytest -m gpu`."""
import json
import math
import os
import random
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent


def _random_actions(env, obs):
    return {k: env.action_space['due' if k.startswith('due') else 'cue'].sample() for k in obs}


def test_python_obs_and_reward_plugins_see_reference_shaped_arguments():
    """Subclasses of the reference-style ABCs run as Python over dict views of the GPU results (d2d_env.py:27-28:
    classes, instantiated with no arguments)."""
    from gym_d2d_amd.envs import D2DEnv
    from gym_d2d_amd.envs.obs_fn import ObsFunction
    from gym_d2d_amd.envs.reward_fn import RewardFunction
    from gym_d2d_amd.spaces import Box
    seen = {}

    class SinrOnlyObs(ObsFunction):
        def get_obs_space(self, env_config):
            return Box(low=-200.0, high=200.0, shape=(2,))

        def get_state(self, actions, state, devices):
            seen['devices'] = len(devices)
            out = {}
            for ids, action in actions.items():
                assert action.tx.id == ids[0] and action.rx.position.as_tuple() == devices[ids[1]].position.as_tuple()
                out[':'.join(ids)] = np.array([state['sinrs_db'][ids], state['snrs_db'][ids]])
            return out

    class MinRateReward(RewardFunction):
        def __call__(self, actions, state):
            worst = min(state['rate_bps'].values())
            assert set(state) >= {'sinrs_db', 'snrs_db', 'rate_bps', 'capacity_mbps'}
            return {':'.join(ids): worst for ids in actions.keys()}

    cfg = {'num_rbs': 5, 'num_cues': 4, 'num_due_pairs': 6, 'obs_fn': SinrOnlyObs, 'reward_fn': MinRateReward}
    env = D2DEnv(cfg)
    assert 'obs_fn' not in cfg and 'reward_fn' not in cfg          # popped from the caller's dict, like the reference
    obs = env.reset()
    assert env.observation_space.shape == (2,) and seen['devices'] == 1 + 4 + 12
    obs, rewards, done, info = env.step(_random_actions(env, obs))
    for key in obs:
        assert obs[key][0] == info[key]['sinr_db'] and obs[key][1] == info[key]['snr_db']
    assert len(set(rewards.values())) == 1 and next(iter(rewards.values())) == min(i['rate_bps'] for i in info.values())
    env.simulator.handle.close()


def test_subclassed_builtin_reward_still_runs_on_the_gpu():
    """SystemCapacityRewardFunction subclass with a different threshold (how SURVEY 8(c) case 11 is built)."""
    from gym_d2d_amd.envs import D2DEnv
    from gym_d2d_amd.envs.reward_fn import SystemCapacityRewardFunction

    class Strict(SystemCapacityRewardFunction):
        def __init__(self):
            super().__init__(min_capacity_mbps=1e9)                 # every non-D2D link fails -> -1 whenever shared

    env = D2DEnv({'num_rbs': 1, 'num_cues': 3, 'num_due_pairs': 3, 'reward_fn': Strict})
    assert env._native_reward
    obs = env.reset()
    obs, rewards, *_ = env.step(_random_actions(env, obs))
    assert set(rewards.values()) == {-1.0}
    env.simulator.handle.close()


def test_device_config_round_trip(tmp_path):
    """save_device_config -> device_config_file (d2d_env.py:124-134; env_config.py:32-37; simulator.py:65-66)."""
    from gym_d2d_amd.envs import D2DEnv
    env = D2DEnv({'num_rbs': 4, 'num_cues': 5, 'num_due_pairs': 5})
    obs = env.reset()
    acts = _random_actions(env, obs)
    first = env.step(acts)
    path = tmp_path / 'device_config.json'
    env.save_device_config(path)
    saved = json.loads(path.read_text())
    assert set(saved) == set(env.simulator.devices) and set(saved['cue00']) == {'position', 'config'}
    saved['due01']['config']['rx_antenna_gain_dBi'] = 6.0           # hand-edit one device, as the README suggests
    path.write_text(json.dumps(saved))
    env2 = D2DEnv({'num_rbs': 4, 'num_cues': 5, 'num_due_pairs': 5, 'device_config_file': path})
    env2.reset()
    for dev_id, dev in env.simulator.devices.items():
        assert env2.simulator.devices[dev_id].position == dev.position
    second = env2.step(acts)
    for key in acts:
        a, b = first[3][key]['sinr_db'], second[3][key]['sinr_db']
        if key == 'due00:due01':
            assert b == pytest.approx(a + 6.0, abs=1e-4)            # +6 dBi at that receiver, signal only
        else:
            assert a == b
    env.simulator.handle.close(); env2.simulator.handle.close()


def test_batched_env_honours_device_config_file(tmp_path):
    """device_config_file with a batch: pinned devices sit at their file position in EVERY env (simulator.py:65-66),
    a DUE receiver whose transmitter is pinned is drawn around the pinned position, per-device overrides apply."""
    from gym_d2d_amd.envs import VecD2DEnv
    cfg_json = {
        'cue01': {'position': [100.0, -50.0], 'config': {'num_subcarriers': 12, 'subcarrier_spacing_kHz': 15,
                                                         'max_tx_power_dBm': 23, 'tx_antenna_gain_dBi': 3.0}},
        'due02': {'position': [-200.0, 25.0], 'config': {'num_subcarriers': 12, 'subcarrier_spacing_kHz': 15,
                                                         'max_tx_power_dBm': 20}},
    }
    path = tmp_path / 'pinned.json'
    path.write_text(json.dumps(cfg_json))
    base = {'num_rbs': 4, 'num_cues': 3, 'num_due_pairs': 3}
    env = VecD2DEnv(dict(base, device_config_file=path), num_envs=64, use_torch=False)
    ref = VecD2DEnv(dict(base), num_envs=64, use_torch=False)
    env.reset(seed=9); ref.reset(seed=9)
    pos, pos_ref = env.simulator.positions(), ref.simulator.positions()
    idx = env.simulator.devices.index
    assert (pos[:, idx['cue01']] == np.float32([100.0, -50.0])).all()
    assert (pos[:, idx['due02']] == np.float32([-200.0, 25.0])).all()
    d = np.hypot(*(pos[:, idx['due03']] - pos[:, idx['due02']]).T)
    assert (d <= 20.0 * (1 + 1e-5)).all() and d.std() > 1.0          # around the PINNED tx, still random per env
    free = [k for name, k in idx.items() if name not in ('cue01', 'due02', 'due03')]
    assert np.array_equal(pos[:, free], pos_ref[:, free])            # everything else: the same counter-based draws
    # +3 dBi on cue01's transmit side: its own SNR rises by exactly 3 dB when placed identically
    ref.simulator.set_positions(pos)
    acts = np.random.default_rng(0).integers(0, 4 * 21, (64, 6)).astype(np.int32)
    a = env.step(acts)[3]['snr_db']; b = ref.step(acts)[3]['snr_db']
    assert np.allclose(a[:, 1] - b[:, 1], 3.0, atol=2e-4) and np.array_equal(a[:, [0, 2, 3, 4, 5]], b[:, [0, 2, 3, 4, 5]])
    env.close(); ref.close()


def test_render_prints_observations(capsys):
    from gym_d2d_amd.envs import D2DEnv
    env = D2DEnv({'num_rbs': 2, 'num_cues': 1, 'num_due_pairs': 1})
    with pytest.raises(AssertionError):
        env.render()
    env.reset()
    env.render()
    assert 'cue00:mbs' in capsys.readouterr().out
    env.simulator.handle.close()


def test_host_reset_reproduces_python_random_stream():
    """random.seed(k) gives the layout the reference would draw (same consumption order: theta then radius; CUEs,
    then per pair tx and the rx rejection loop - simulator.py:61-75) - bit for bit, in float64: nothing is rounded."""
    from gym_d2d_amd.envs import D2DEnv
    env = D2DEnv({'num_cues': 3, 'num_due_pairs': 3})
    random.seed(12)
    env.reset()
    got = np.array([d.position.as_tuple() for d in env.simulator.devices.values()])
    random.seed(12)

    def draw(radius):
        t = 2 * math.pi * random.random()
        r = radius * math.sqrt(random.random())
        return r * math.cos(t), r * math.sin(t)

    want = [(0.0, 0.0)] + [draw(500.0) for _ in range(3)]
    for _ in range(3):
        tx = draw(500.0)
        want.append(tx)
        while True:
            dx, dy = draw(20.0)
            x, y = tx[0] + dx, tx[1] + dy
            if not x * x + y * y > 500.0 ** 2:
                want.append((x, y))
                break
    assert np.array_equal(got, np.array(want, dtype=np.float64))
    assert (got != got.astype(np.float32)).any()
    env.simulator.handle.close()


@pytest.mark.parametrize('script', ['simple_env.py', 'custom_path_loss.py', 'array_path_loss.py', 'saving_loading_device_config.py', 'vec_env.py'])
def test_examples_run(script):
    r = subprocess.run([sys.executable, str(ROOT / 'examples' / script)], capture_output=True, text=True, timeout=600,
                       env={**os.environ, 'PYTHONPATH': str(ROOT)})
    assert r.returncode == 0, r.stderr[-2000:]
    assert r.stdout.strip()


def test_integration_md_binding_stub_runs_as_written():
    """INTEGRATION.md section B shows the ctypes binding a maintainer of the reference would add (`gym_d2d/_hip.py`).  This
    test takes that code block OUT OF THE DOCUMENT, points it at the in-tree library, drives it with reference-shaped objects
    (devices, config, path loss, Actions - this package's mirrors expose the same attributes) and checks the `state` dict it
    returns against the oracle: the documented binding is executable, not prose."""
    import re
    from gym_d2d_amd import _native
    from gym_d2d_amd.actions import Action, Actions
    from gym_d2d_amd.envs.env_config import EnvConfig
    from gym_d2d_amd.link_type import LinkType
    from gym_d2d_amd.path_loss import LogDistancePathLoss
    from gym_d2d_amd.position import Position
    from gym_d2d_amd.simulator import create_devices
    from oracle import d2d_oracle as orc
    text = (ROOT / 'INTEGRATION.md').read_text()
    block = re.search(r'```python\n(# gym_d2d/_hip\.py.*?)```', text, re.S).group(1)
    block = block.replace("C.CDLL('libd2d_hip.so')", f"C.CDLL({str(_native.LIB_PATH)!r})")
    ns = {}
    exec(compile(block, 'INTEGRATION.md::_hip.py', 'exec'), ns)
    config = EnvConfig(num_rbs=6, num_cues=5, num_due_pairs=7)
    devices = create_devices(config)
    rng = np.random.default_rng(8)
    for d in devices.values():
        if d.id != 'mbs':
            r, th = 400.0 * math.sqrt(rng.random()), 2 * math.pi * rng.random()
            d.set_position(Position(r * math.cos(th), r * math.sin(th)))      # Python floats, as the reference's reset leaves them
    for tx, rx in devices.dues.values():                          # receivers near their transmitters (one of them 0.2 m away)
        rx.set_position(Position(tx.position.x + (0.15 if tx.id == 'due00' else 7.0), tx.position.y - (0.1 if tx.id == 'due00' else 5.0)))
    core = ns['HipSimulatorCore'](config, devices, LogDistancePathLoss(config.carrier_freq_GHz))
    core.push_positions()
    acts = Actions()
    for k, (cid, cue) in enumerate(devices.cues.items()):
        acts[(cid, 'mbs')] = Action(cue, devices.bs, LinkType.UPLINK, k % 6, 10 + k)
    for k, ((t, r), (tx, rx)) in enumerate(devices.dues.items()):
        acts[(t, r)] = Action(tx, rx, LinkType.SIDELINK, (2 * k) % 6, 3 + k)
    state = core.step(acts)
    assert set(state) == {'sinrs_db', 'snrs_db', 'rate_bps', 'capacity_mbps'} and list(state['sinrs_db']) == list(acts.keys())
    ids, cfgs, is_bs = orc.device_configs(5, 7)
    index = {d: k for k, d in enumerate(devices.keys())}
    pos = np.array([d.position.as_tuple() for d in devices.values()], dtype=np.float64)[None]
    tx_i = np.array([index[t] for t, _ in acts.keys()]); rx_i = np.array([index[r] for _, r in acts.keys()])
    rb = np.array([[a.rb for a in acts.values()]]); pw = np.array([[a.tx_pwr_dBm for a in acts.values()]])
    ref = orc.step(pos, tx_i, rx_i, rb, pw, orc.device_columns(cfgs, is_bs), orc.PathLossSpec())
    for key, f in (('sinrs_db', 'sinr_db'), ('snrs_db', 'snr_db'), ('rate_bps', 'rate_bps'), ('capacity_mbps', 'capacity_mbps')):
        got = np.array(list(state[key].values()))
        assert np.max(np.abs(got - ref[f][0]) / np.maximum(np.abs(ref[f][0]), 1.0)) <= 1e-5, key
    ns['lib'].d2d_destroy(core.h)
