"""CPU-only tests of the host side: config surface, device tables, path-loss lowering, C-ABI export list.
No compute call is made (there is no GPU here and no CPU fallback to call)."""
import json
import math
import random
import re
from pathlib import Path

import numpy as np
import pytest

from golden_util import GOLDEN_DIR, known_answers, load_case
from oracle import d2d_oracle as orc

ROOT = Path(__file__).resolve().parent.parent


def test_library_exports_every_symbol_of_the_header():
    from gym_d2d_amd import _native
    lib = _native.load_library()                       # resolves every name in SIGNATURES or raises
    header = (ROOT / 'include' / 'd2d_hip.h').read_text()
    declared = set(re.findall(r'^(?:int|const char\*)\s+(d2d_\w+)\s*\(', header, flags=re.M))
    assert declared, 'header parse failed'
    assert declared == set(_native.SIGNATURES), declared ^ set(_native.SIGNATURES)
    for name in declared:
        assert getattr(lib, name) is not None
    assert lib.d2d_abi_version() == _native.ABI_VERSION


def test_product_path_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    from gym_d2d_amd import _native
    from gym_d2d_amd.envs import D2DEnv
    with pytest.raises(_native.NativeError):
        D2DEnv({})


def test_gym_make_registration_is_executed(tmp_path):
    """gym_d2d/__init__.py:8-11: `register(id='D2DEnv-v0', entry_point=...:D2DEnv)` at import, `gym.make('D2DEnv-v0',
    env_config=...)` as the factory.  gym is not installed here: a child process gets a stand-in `gym` on its path, imports
    this package (which must then take its gym branch) and calls gym.make - which reaches D2DEnv.__init__: a working env on a
    GPU box (tests/test_gpu_round3.py runs the same child there), the loud NativeError without one."""
    from gym_stub_util import run_gym_make
    out = run_gym_make(tmp_path, ROOT)
    assert out['have_gym'] is True and out['space_is_gym'] is True
    assert out['entry_point'] == 'gym_d2d_amd.envs:D2DEnv'
    import torch
    assert out['make'] == ('ok' if torch.cuda.is_available() else 'NativeError')


def test_product_package_never_imports_the_oracle():
    for path in (ROOT / 'gym_d2d_amd').rglob('*.py'):
        text = path.read_text()
        assert 'import oracle' not in text and 'from oracle' not in text, path


def test_env_config_surface():
    from gym_d2d_amd.envs.env_config import EnvConfig
    from gym_d2d_amd.path_loss import LogDistancePathLoss
    from gym_d2d_amd.traffic_model import UplinkTrafficModel
    c = EnvConfig()
    assert (c.num_rbs, c.num_cues, c.num_due_pairs, c.cell_radius_m, c.d2d_radius_m) == (25, 25, 25, 500.0, 20.0)
    assert (c.due_min_tx_power_dBm, c.due_max_tx_power_dBm, c.cue_max_tx_power_dBm, c.mbs_max_tx_power_dBm) == (0, 20, 23, 46)
    assert c.path_loss_model is LogDistancePathLoss and c.traffic_model is UplinkTrafficModel
    assert (c.carrier_freq_GHz, c.num_subcarriers, c.subcarrier_spacing_kHz, c.channel_bandwidth_MHz) == (2.1, 12, 15, 20.0)
    assert c.device_config_file is None and c.devices == {}
    assert c.num_pwr_actions == {'due': 21, 'cue': 24, 'mbs': 47}          # 525/600/1175 actions at 25 RBs
    with pytest.raises(TypeError):
        EnvConfig(no_such_key=1)


def test_create_devices_ids_and_configs():
    """Mirrors the reference's test_simulator.py:5-18."""
    from gym_d2d_amd.envs.env_config import EnvConfig
    from gym_d2d_amd.simulator import create_devices
    devs = create_devices(EnvConfig(num_cues=3, num_due_pairs=2, cue_max_tx_power_dBm=9, due_max_tx_power_dBm=4))
    assert list(devs) == ['mbs', 'cue00', 'cue01', 'cue02', 'due00', 'due01', 'due02', 'due03']
    assert len(devs.cues) == 3 and len(devs.dues) == 2
    assert ('due00', 'due01') in devs.dues and devs.due_pairs['due02'] == 'due03' and devs.due_pairs_inv['due01'] == 'due00'
    assert devs['cue00'].max_tx_power_dBm == 9 and devs['due03'].max_tx_power_dBm == 4
    assert devs.index_of('due02') == 6
    with pytest.raises(KeyError):
        devs['nobody']


def test_device_defaults_and_eirp():
    """Mirrors the reference's test_device.py:71-99."""
    from gym_d2d_amd.device import BaseStation, UserEquipment
    kat = known_answers()
    ue, bs = UserEquipment('ue'), BaseStation('bs')
    assert ue.eirp_dBm(12) == kat['test_device.py:73-77 ue eirp(12)']
    assert bs.eirp_dBm(46) == kat['test_device.py:80-85 bs eirp(46)']
    assert ue.rx_sensitivity_dBm == -107.5 and ue.rx_noise_floor_dBm == -97.5
    assert bs.rx_sensitivity_dBm == pytest.approx(-123.4)
    assert ue.rb_bandwidth_kHz == 180 and isinstance(ue.rb_bandwidth_kHz, int)
    assert ue.rx_signal_level_dBm(10.0, 100.0) == 10.0 - 100.0 + 0.0 - 3.0
    assert bs.rx_signal_level_dBm(10.0, 100.0) == 10.0 - 100.0 + 17.5 - 2.0 + 2.0
    over = UserEquipment('x', {'body_loss_dB': 1.0, 'extra': 5})
    assert over.body_loss_dB == 1.0 and over.config['extra'] == 5 and over.ix_margin_dB == 3.0


@pytest.mark.parametrize('name', ['case01_default', 'case07_device_config'])
def test_link_budget_columns_match_oracle_and_reference_configs(name):
    from gym_d2d_amd.device import link_budget_columns
    from gym_d2d_amd.envs.env_config import EnvConfig
    from gym_d2d_amd.simulator import create_devices
    case = load_case(name)
    m = case.meta
    kw = dict(num_rbs=m['num_rbs'], num_cues=m['num_cues'], num_due_pairs=m['num_due_pairs'])
    if '07' in name:
        kw['device_config_file'] = GOLDEN_DIR / 'case07_device_config.json'
    devs = create_devices(EnvConfig(**kw))
    assert [str(i) for i in devs] == case.ids
    for dev, ref_cfg in zip(devs.values(), case.cfgs):
        assert dev.config == ref_cfg                                   # merged config == the reference's
    cols = link_budget_columns(devs.values())
    ref = orc.device_columns(case.cfgs, case.is_bs)
    for a, b in (('eirp_off_db', ref.eirp_off_db), ('rx_off_db', ref.rx_off_db), ('noise_dbm', ref.noise_dbm),
                 ('sens_dbm', ref.sens_dbm), ('bw_hz', ref.bw_hz)):
        assert np.allclose(cols[a], b, rtol=0, atol=1e-12), a


def test_path_loss_known_answers_and_lowering():
    """Reference pins test_path_loss.py:8-27,42-52, and: the power-law columns reproduce __call__ to 1e-12."""
    from gym_d2d_amd.device import BaseStation, UserEquipment
    from gym_d2d_amd.path_loss import (AreaType, CostHataPathLoss, FreeSpacePathLoss, LogDistancePathLoss,
                                       ShadowingPathLoss, pl_constant_dB)
    from gym_d2d_amd.position import Position
    kat = known_answers()
    assert pl_constant_dB(2.1, 2.0) == pytest.approx(kat['test_path_loss.py:11 pl_constant_dB(2.1,2.0)'], abs=1e-12)
    bs, ue = BaseStation('mbs'), UserEquipment('ue')
    bs.set_position(Position(0, 0))
    ld = LogDistancePathLoss(2.1)
    hu = CostHataPathLoss(2.1, AreaType.URBAN)
    for d, k_ld, k_bu, k_ub in ((250.0, 'test_path_loss.py:25 logdist 2.1GHz 250m', 'test_path_loss.py:48 hata urban bs->ue 250m',
                                 'test_path_loss.py:49 hata urban ue->bs 250m'),
                                (500.0, 'test_path_loss.py:27 logdist 2.1GHz 500m', 'test_path_loss.py:51 hata urban bs->ue 500m',
                                 'test_path_loss.py:52 hata urban ue->bs 500m')):
        ue.set_position(Position(0, d))
        assert ld(bs, ue) == pytest.approx(kat[k_ld], rel=1e-6)
        assert hu(bs, ue) == pytest.approx(kat[k_bu], abs=1e-10)
        assert hu(ue, bs) == pytest.approx(kat[k_ub], abs=1e-10)
    assert FreeSpacePathLoss(2.1)(bs, ue) == ld(bs, ue)
    # lowering: a_tx[tx] + a_rx[rx] + 10 n[tx] log10(d) == model(tx, rx)
    ue.set_position(Position(123.0, -77.0))
    for model in (ld, LogDistancePathLoss(2.1, ple=3.5), hu, CostHataPathLoss(2.1), CostHataPathLoss(0.15, AreaType.URBAN)):
        cols = model.power_law_columns([bs, ue])
        d = bs.position.distance(ue.position)
        for t, r, a, b in ((bs, ue, 0, 1), (ue, bs, 1, 0)):
            lowered = cols['a_tx_db'][a] + cols['a_rx_db'][b] + 10 * cols['exponent'][a] * math.log10(d)
            assert lowered == pytest.approx(model(t, r), abs=1e-10)
    # subclass that changes the formula falls back to the table route
    class Foo(LogDistancePathLoss):
        def __call__(self, tx, rx):
            return 1.0
    assert Foo(2.1).power_law_columns([bs, ue]) is None
    tab = Foo(2.1).table_db([bs, ue])
    assert tab[0, 1] == 1.0 and np.isnan(tab[0, 0])
    # shadowing: d <= d0 is deterministic log-distance; beyond it the draw has std chi
    sh = ShadowingPathLoss(2.1, d0_m=100.0, chi_dB=2.7)
    ue.set_position(Position(50.0, 0))
    assert sh(bs, ue) == ld(bs, ue)
    ue.set_position(Position(400.0, 0))
    random.seed(0)
    draws = np.array([sh(bs, ue) for _ in range(4000)])
    assert abs(draws.mean() - ld(bs, ue)) < 0.2 and abs(draws.std() - 2.7) < 0.15


def test_conversion_known_answers():
    """Mirrors the reference's test_conversion.py:6-42."""
    from gym_d2d_amd.conversion import W_to_dBm, dB_to_linear, dBm_to_W, linear_to_dB
    kat = known_answers()
    assert dB_to_linear(1) == pytest.approx(kat['test_conversion.py:8 dB_to_linear(1)'], rel=1e-6)
    assert linear_to_dB(2) == pytest.approx(kat['test_conversion.py:17 linear_to_dB(2)'], rel=1e-5)
    assert dBm_to_W(30) == pytest.approx(kat['test_conversion.py:31 dBm_to_W(30)'])
    assert W_to_dBm(0.2) == pytest.approx(kat['test_conversion.py:37 W_to_dBm(0.2)'], rel=1e-6)
    assert dB_to_linear(0) == 1.0 and linear_to_dB(1) == 0.0


def test_position_samplers():
    """Mirrors the reference's test_position.py:12-44 + same RNG consumption order as position.py."""
    from gym_d2d_amd.position import Position, get_random_position, get_random_position_nearby
    a, b = Position(1.0, 2.0), Position(4.0, 6.0)
    assert a.distance(b) == b.distance(a) == 5.0 and a.as_tuple() == (1.0, 2.0)
    for seed in range(10):
        random.seed(seed)
        p = get_random_position(500.0)
        assert p.distance(Position(0, 0)) <= 500.0
        q = get_random_position_nearby(500.0, p, 20.0)
        assert p.distance(q) <= 20.0 + 1e-9 and q.distance(Position(0, 0)) <= 500.0
        random.seed(seed)                          # theta first, then radius - the reference's order
        theta = 2 * math.pi * random.random(); r = 500.0 * math.sqrt(random.random())
        assert p.as_tuple() == (r * math.cos(theta), r * math.sin(theta))


def test_actions_rb_index():
    """Mirrors the reference's test_actions.py:24-48."""
    from gym_d2d_amd.actions import Action, Actions
    from gym_d2d_amd.device import BaseStation, UserEquipment
    from gym_d2d_amd.link_type import LinkType
    bs, c0, c1, d0, d1 = BaseStation('mbs'), UserEquipment('cue00'), UserEquipment('cue01'), UserEquipment('due00'), UserEquipment('due01')
    a = Action(c0, bs, LinkType.UPLINK, 0, 23); b = Action(c1, bs, LinkType.UPLINK, 1, 23); c = Action(d0, d1, LinkType.SIDELINK, 0, 10)
    acts = Actions({('cue00', 'mbs'): a, ('cue01', 'mbs'): b, ('due00', 'due01'): c})
    assert acts.get_actions_by_rb(0) == {a, c} and acts.get_actions_by_rb(1) == {b} and acts.get_actions_by_rb(9) == set()
    acts.clear()
    assert len(acts) == 0 and acts.get_actions_by_rb(0) == set()
    assert LinkType.UPLINK.value == 1 and LinkType.DOWNLINK.value == 2 and LinkType.SIDELINK.value == 3


def test_traffic_models():
    from gym_d2d_amd.envs.env_config import EnvConfig
    from gym_d2d_amd.link_type import LinkType
    from gym_d2d_amd.simulator import create_devices
    from gym_d2d_amd.traffic_model import DownlinkTrafficModel, UplinkTrafficModel
    devs = create_devices(EnvConfig(num_cues=5, num_due_pairs=1))
    up = UplinkTrafficModel(3).get_traffic(devs)
    assert list(up.keys()) == [(f'cue0{i}', 'mbs') for i in range(5)]
    assert [a.rb for a in up.values()] == [0, 1, 2, 0, 1] and all(a.tx_pwr_dBm == 23 and a.link_type == LinkType.UPLINK for a in up.values())
    down = DownlinkTrafficModel(3).get_traffic(devs)
    assert list(down.keys())[0] == ('mbs', 'cue00') and all(a.link_type == LinkType.DOWNLINK for a in down.values())
    rb, pwr = UplinkTrafficModel(3).assignments(devs)
    assert list(rb) == [0, 1, 2, 0, 1] and list(pwr) == [23] * 5


def test_spaces_and_obs_space():
    from gym_d2d_amd.envs.env_config import EnvConfig
    from gym_d2d_amd.envs.obs_fn import LinearObsFunction
    from gym_d2d_amd.spaces import Dict, Discrete
    d = Dict({'due': Discrete(525), 'cue': Discrete(600)})
    assert d['due'].n == 525 and 0 <= d['cue'].sample() < 600
    box = LinearObsFunction().get_obs_space(EnvConfig())
    assert box.shape == (300,) and float(np.max(box.high)) == 500.0 and float(np.min(box.low)) == -500.0


def test_builtin_plugins_refuse_host_evaluation():
    """There is no CPU implementation of the built-in reward / obs functions to fall back on."""
    from gym_d2d_amd.actions import Actions
    from gym_d2d_amd.envs.obs_fn import LinearObsFunction
    from gym_d2d_amd.envs.reward_fn import SystemCapacityRewardFunction
    with pytest.raises(RuntimeError):
        LinearObsFunction().get_state(Actions(), {'sinrs_db': {}}, None)
    with pytest.raises(RuntimeError):
        SystemCapacityRewardFunction()(Actions(), {'capacity_mbps': {}})


def test_merge_dicts():
    from gym_d2d_amd.utils import merge_dicts
    a = {'x': 1, 'n': {'p': 1, 'q': 2}}
    out = merge_dicts(a, {'x': 2, 'n': {'q': 3, 'r': 4}, 'y': 5})
    assert out is a and a == {'x': 2, 'n': {'p': 1, 'q': 3, 'r': 4}, 'y': 5}


def test_philox_known_answers():
    """Random123 kat_vectors for philox4x32-10."""
    f = lambda *a: [int(x) for x in orc.philox4x32_10(*a)]
    assert f(0, 0, 0, 0, 0, 0) == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    assert f(*([0xffffffff] * 6)) == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    assert f(0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344, 0xa4093822, 0x299f31d0) == \
        [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]


def test_reset_action_stream_is_shard_invariant_and_backend_independent():
    """The random actions of VecD2DEnv.reset (d2d_env.py:54-60) are a pure function of (seed, episode, GLOBAL env, column):
    NumPy and torch produce the same integers, and any split of the env axis reproduces its slice of the whole batch."""
    import torch
    from gym_d2d_amd.envs import _rng
    highs = [25 * 24] * 3 + [25 * 21] * 4
    whole = _rng.uniform_ints_numpy(1234, 3, 0, 64, 7, highs)
    assert whole.dtype == np.int32 and whole.shape == (64, 7)
    assert (whole >= 0).all() and (whole < np.array(highs)[None, :]).all()
    assert np.array_equal(_rng.uniform_ints_torch(torch, 1234, 3, 0, 64, 7, highs, 'cpu').numpy(), whole)
    for first, count in ((0, 16), (16, 16), (40, 24)):
        assert np.array_equal(_rng.uniform_ints_numpy(1234, 3, first, count, 7, highs), whole[first:first + count])
    assert not np.array_equal(_rng.uniform_ints_numpy(1234, 4, 0, 64, 7, highs), whole)          # next episode: new draws
    assert not np.array_equal(_rng.uniform_ints_numpy(1235, 3, 0, 64, 7, highs), whole)
    big = _rng.uniform_ints_numpy(7, 0, 0, 2048, 64, 256 * 21)
    assert abs(big.mean() / (256 * 21) - 0.5) < 0.01 and len(np.unique(big)) > 5000


def test_make_action_equals_the_dataclass_constructor():
    """actions.make_action is Action(...) without the frozen dataclass's five object.__setattr__ calls: same fields, same
    equality / hash (Actions.get_actions_by_rb puts them in sets), still frozen."""
    import dataclasses
    from gym_d2d_amd.actions import Action, make_action
    from gym_d2d_amd.device import BaseStation, UserEquipment
    from gym_d2d_amd.id import Id
    from gym_d2d_amd.link_type import LinkType
    tx, rx = UserEquipment(Id('cue00'), {}), BaseStation(Id('mbs'), {})
    fast, slow = make_action(tx, rx, LinkType.UPLINK, 3, 17), Action(tx, rx, LinkType.UPLINK, 3, 17)
    assert fast == slow and hash(fast) == hash(slow) and dataclasses.asdict(fast).keys() == dataclasses.asdict(slow).keys()
    assert (fast.tx, fast.rx, fast.link_type, fast.rb, fast.tx_pwr_dBm) == (tx, rx, LinkType.UPLINK, 3, 17)
    with pytest.raises(dataclasses.FrozenInstanceError):
        fast.rb = 4


def test_array_path_loss_derives_the_per_object_call_from_compute():
    """ArrayPathLoss (round 6): one definition - compute(view) on arrays - also answers the reference's per-object contract
    model(tx, rx) -> dB (path_loss.py:12-25), which the single-env D2DEnv and PathLoss.table_db use; a zero distance raises what
    math.log10(0) raises in a per-object model."""
    import math
    from gym_d2d_amd.envs.env_config import EnvConfig
    from gym_d2d_amd.path_loss import ArrayPathLoss, PathLoss, PathLossView
    from gym_d2d_amd.position import Position
    from gym_d2d_amd.simulator import create_devices

    class Arr(ArrayPathLoss):
        def compute(self, view):
            d = view.distance()
            return 20 * view.xp.log10(d) + 38.0 - view.tx_column(lambda t: t.tx_antenna_gain_dBi) - 0.5 * view.rx_column(lambda r: r.antenna_height_m)

    class Obj(PathLoss):
        def __call__(self, tx, rx):
            return 20 * math.log10(tx.position.distance(rx.position)) + 38.0 - tx.tx_antenna_gain_dBi - 0.5 * rx.antenna_height_m
    devs = list(create_devices(EnvConfig(num_cues=3, num_due_pairs=2)).values())
    rng = np.random.default_rng(2)
    for d in devs[1:]:
        d.set_position(Position(*rng.uniform(-400, 400, 2)))
    a, o = Arr(2.1), Obj(2.1)
    for tx in devs:
        for rx in devs:
            if tx is not rx:
                assert abs(a(tx, rx) - o(tx, rx)) < 1e-12
    with pytest.raises(ValueError, match='math domain error'):
        a(devs[1], devs[1])
    ta, to = a.table_db(devs, [1, 2], [0, 3]), o.table_db(devs, [1, 2], [0, 3])
    assert np.array_equal(np.isnan(ta), np.isnan(to)) and np.allclose(ta[~np.isnan(ta)], to[~np.isnan(to)], rtol=0, atol=1e-12)
    # the view: [B, N, N] by (tx link j, rx link i), float64 differences of float32 coordinates
    x = np.float32([[0.0, 3.0], [10.0, 10.0]])
    view = PathLossView(np, x, x * 0, x[:, ::-1], x * 0 + 4.0, devs[:2], devs[2:4])
    dist = view.distance()
    assert dist.shape == (2, 2, 2) and dist.dtype == np.float64 and dist[0, 0, 0] == 5.0 and dist[0, 0, 1] == 4.0 and dist[1, 1, 0] == 4.0
    assert view.tx_column(lambda t: 1.5).shape == (1, 2, 1) and view.rx_column(lambda r: 2.5).shape == (1, 1, 2)


def test_handle_routes_float64_positions_to_the_float64_entry(monkeypatch):
    """Handle.set_positions: float64 arrays go through d2d_set_positions_f64 (the reference's precision, position.py:7-12), anything
    else through the float32 entry - decided by dtype alone, without a GPU."""
    import ctypes as C
    from gym_d2d_amd import _native
    calls = []

    class Lib:
        def d2d_set_positions(self, h, x, y, b, n): calls.append(('f32', b, n)); return 0
        def d2d_set_positions_f64(self, h, x, y, b, n): calls.append(('f64', b, n)); return 0
    h = _native.Handle.__new__(_native.Handle)
    h._lib, h._h, h.num_devices = Lib(), C.c_void_p(1), 3
    h.set_positions(np.zeros((2, 3)), np.zeros((2, 3)))
    h.set_positions(np.zeros((2, 3), np.float32), np.zeros((2, 3), np.float32), env_begin=1)
    h.set_positions(np.zeros((2, 3)), np.zeros((2, 3), np.float32))          # mixed: float32
    assert calls == [('f64', 0, 2), ('f32', 1, 2), ('f32', 0, 2)]
    with pytest.raises(ValueError):
        h.set_positions(np.zeros((2, 4)), np.zeros((2, 4)))
    h._h = None
