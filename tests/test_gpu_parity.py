"""GPU parity tests: the HIP path (through the C ABI) against the golden vectors captured from the reference and
against the NumPy fp64 oracle on the same inputs.  Run on the MI355X box with `pytest -m gpu`.

Parity metric: max |d| / max(|ref|, 1) <= 1e-5 for floating outputs; bit-exact for the
integer decode and for copied positions.
"""
import numpy as np
import pytest

from golden_util import case_names, load_case, rel_err
from oracle import d2d_oracle as orc
from sim_util import default_links, env_config_for, oracle_spec, random_layout

pytestmark = pytest.mark.gpu
TOL = 1e-5


@pytest.fixture(scope='module')
def native():
    from gym_d2d_amd import _native
    _native.load_library()
    return _native


def _sim_for(case):
    from gym_d2d_amd.simulator import Simulator
    sim = Simulator(env_config_for(case))
    assert [str(i) for i in sim.devices.keys()] == case.ids
    # the `unrounded` cases hold the reference's own float64 layouts: they go up as they are (d2d_set_positions_f64)
    sim.set_positions(case.pos[None] if 'unrounded' in case.name else case.pos[None].astype(np.float32))
    return sim


def _check_step(sim, native, case, s, tag):
    for f, buf in (('sinr_db', native.BUF_SINR_DB), ('snr_db', native.BUF_SNR_DB), ('rate_bps', native.BUF_RATE_BPS),
                   ('capacity_mbps', native.BUF_CAPACITY)):
        got = sim.fetch(buf)[0]
        assert rel_err(got, getattr(s, f)) <= TOL, (tag, f, rel_err(got, getattr(s, f)))
    table = sim.fetch(native.BUF_OBS_TABLE)[0]
    assert rel_err(table, s.obs_table) <= TOL, tag
    assert (table[:, :4] == s.obs_table[:, :4].astype(np.float32)).all(), 'positions are copies: exact'
    obs = sim.fetch(native.BUF_OBS)[0]
    n = len(s.keys)
    assert obs.shape == (n, 6 * n)
    assert rel_err(obs[s.obs_rows], s.obs) <= TOL, tag
    # whole expansion against the oracle's expansion of the kernel's own table: pure data movement, bit exact
    assert (obs == orc.expand_obs(table[None])[0]).all(), tag


@pytest.mark.parametrize('walk', [-1, 2], ids=['auto', 'member_lists'])
@pytest.mark.parametrize('name', [n for n in case_names() if 'shadowing' not in n])
def test_golden_cases_low_level(native, name, walk):
    """Every golden case through Simulator.step_arrays (C ABI), all three reward functions; with the default interferer
    search and with the per-RB member lists (the reference's 4-RB / 40-link case puts ten links on every RB: list overflow)."""
    case = load_case(name)
    sim = _sim_for(case)
    h = sim.handle
    h.set_obs_mode(native.OBS_LINEAR)
    h.set_tuning(native.TUNE_STEP_WALK, walk)
    for k, s in enumerate(case.steps):
        keys = [tuple(key.split(':')) for key in s.keys]
        sim.set_links(keys)
        tag = (name, k)
        use_raw = hasattr(s, 'raw') and (s.raw >= 0).all()
        for rid, param, field in ((native.REWARD_SYSTEM_CAPACITY, 0.0, 'reward_system_capacity'),
                                  (native.REWARD_SHANNON, -70.0, 'reward_shannon'),
                                  (native.REWARD_CUE_SINR_SHANNON, 0.0, 'reward_cue_sinr_shannon')):
            h.set_reward(rid, param)
            if use_raw:
                sim.step_arrays(s.raw[None])
                assert (sim.fetch(native.BUF_RB)[0] == s.rb).all(), tag          # integer decode: bit exact
                assert (sim.fetch(native.BUF_PWR)[0] == s.pwr).all(), tag
            else:
                sim.step_arrays(rb=s.rb[None], pwr=s.pwr[None])
            assert sim.check_flags() & native.FLAG_ZERO_DISTANCE == 0
            assert rel_err(sim.fetch(native.BUF_REWARD)[0], getattr(s, field)) <= TOL, (tag, field)
        _check_step(sim, native, case, s, tag)
        for f in vars(s):
            if f.startswith('reward_system_capacity_min'):
                m = float(f[len('reward_system_capacity_min'):].replace('p', '.'))
                h.set_reward(native.REWARD_SYSTEM_CAPACITY, m)
                sim.step_arrays(rb=s.rb[None], pwr=s.pwr[None])
                assert rel_err(sim.fetch(native.BUF_REWARD)[0], getattr(s, f)) <= TOL, (tag, f)
    h.close()


@pytest.mark.parametrize('name', ['case01_default', 'case05_due_subset', 'case06_downlink', 'case07_device_config',
                                  'case10_custom_pl', 'case12_array_actions', 'case16_unrounded_default',
                                  'case16_unrounded_device_config'])
def test_golden_cases_through_d2d_env(native, name):
    """The drop-in D2DEnv (dict in / dict out) replays the reference's episodes."""
    from gym_d2d_amd.envs import D2DEnv
    from gym_d2d_amd.position import Position
    case = load_case(name)
    env = D2DEnv(env_config_for(case))
    env.reset()
    for dev, xy in zip(env.simulator.devices.values(), case.pos):
        dev.set_position(Position(float(xy[0]), float(xy[1])))
    env.simulator.push_positions()
    if name.endswith('device_config'):
        import json
        from golden_util import GOLDEN_DIR
        pinned = json.loads((GOLDEN_DIR / f'{name}.json').read_text())
        env.reset()     # pinned devices must land on their file positions (simulator.py:65-66), in the file's own precision
        for dev_id, entry in pinned.items():
            if dev_id != 'mbs':
                assert env.simulator.devices[dev_id].position.as_tuple() == tuple(entry['position'])
        for dev, xy in zip(env.simulator.devices.values(), case.pos):
            dev.set_position(Position(float(xy[0]), float(xy[1])))
        env.simulator.push_positions()
    env.num_steps = 0
    for k, s in enumerate(case.steps[1:], start=1):
        if hasattr(s, 'raw_rb_pwr'):
            raw = {key: np.array([[r], [p]]) for key, (r, p) in zip(s.keys, s.raw_rb_pwr)}
        else:
            raw = {key: int(a) for key, a in zip(s.keys, s.raw)}
        obs, rewards, game_over, info = env.step(raw)
        assert list(obs.keys()) == s.keys and list(rewards.keys()) == s.keys and list(info.keys()) == s.keys
        assert game_over == {'__all__': bool(s.game_over)}
        n = len(s.keys)
        for row, i in zip(s.obs, s.obs_rows):
            got = obs[s.keys[i]]
            assert got.shape == (6 * n,) and got.dtype == np.float64
            assert rel_err(got, row) <= TOL
        assert rel_err([rewards[key] for key in s.keys], s.reward_env) <= TOL
        for f, g in (('sinr_db', 'sinr_db'), ('snr_db', 'snr_db'), ('rate_bps', 'rate_bps'),
                     ('capacity_mbps', 'capacity_mbps')):
            assert rel_err([info[key][f] for key in s.keys], getattr(s, g)) <= TOL
        assert [info[key]['rb'] for key in s.keys] == list(s.rb)
        assert [info[key]['tx_pwr_dbm'] for key in s.keys] == list(s.pwr)
        assert all(isinstance(info[key]['rb'], int) and isinstance(info[key]['sinr_db'], float) for key in s.keys)
    env.simulator.handle.close()


def _batch(native, num_envs, rbs, cues, dues, seed, **cfg):
    from gym_d2d_amd.simulator import Simulator
    rng = np.random.default_rng(seed)
    sim = Simulator(dict(num_rbs=rbs, num_cues=cues, num_due_pairs=dues, num_envs=num_envs, **cfg),
                    max_links=cues + dues)
    pos = random_layout(rng, num_envs, cues, dues)
    sim.set_positions(pos)
    sim.set_links(sim.default_link_keys())
    p = sim.config.num_pwr_actions
    raw = np.concatenate([rng.integers(0, rbs * p['cue'], (num_envs, cues)),
                          rng.integers(0, rbs * p['due'], (num_envs, dues))], axis=1).astype(np.int32)
    return sim, pos, raw


def _oracle_batch(sim, pos, raw, min_cap=0.0, with_obs=False):
    ids, cfgs, is_bs = orc.device_configs(sim.config.num_cues, sim.config.num_due_pairs)
    cols = orc.device_columns(cfgs, is_bs)
    tx, rx, ty = default_links(sim.config.num_cues, sim.config.num_due_pairs)
    assert (tx == sim.link_tx).all() and (rx == sim.link_rx).all() and (ty == sim.link_type).all()
    return orc.full_step(pos.astype(np.float64), tx, rx, ty, raw, cols, orc.PathLossSpec(), with_obs=with_obs,
                         min_capacity_mbps=min_cap, chunk=16)


@pytest.mark.parametrize('shape', [(64, 25, 25, 25), (16, 256, 256, 256), (32, 4, 10, 30), (8, 3, 0, 7), (8, 5, 9, 0),
                                   (4, 1, 33, 32)])
def test_random_batches_match_oracle(native, shape):
    b, rbs, cues, dues = shape
    sim, pos, raw = _batch(native, b, rbs, cues, dues, seed=sum(shape))
    sim.handle.set_obs_mode(native.OBS_LINEAR)
    sim.step_arrays(raw)
    assert sim.check_flags() == 0
    ref = _oracle_batch(sim, pos, raw, with_obs=True)
    for f, buf in (('sinr_db', native.BUF_SINR_DB), ('snr_db', native.BUF_SNR_DB), ('rate_bps', native.BUF_RATE_BPS),
                   ('capacity_mbps', native.BUF_CAPACITY)):
        assert rel_err(sim.fetch(buf), ref[f]) <= TOL, f
    assert (sim.fetch(native.BUF_RB) == ref['rb']).all() and (sim.fetch(native.BUF_PWR) == ref['pwr']).all()
    assert rel_err(sim.fetch(native.BUF_REWARD), np.repeat(ref['reward'][:, None], cues + dues, 1)) <= TOL
    assert rel_err(sim.fetch(native.BUF_OBS), ref['obs']) <= TOL
    sim.handle.close()


@pytest.mark.parametrize('shape', [(2, 64, 1024, 1024), (2, 8, 1024, 1024), (3, 1, 1, 0), (3, 1, 0, 1), (1, 2000, 3, 3)])
def test_extreme_shapes(native, shape):
    """Maximum links per env (2048: 90 KB of LDS staging; with 64 RBs the mask table no longer fits and the kernel
    takes the all-pairs path, with 8 RBs it fits), single-link envs, far more RBs than links."""
    b, rbs, cues, dues = shape
    sim, pos, raw = _batch(native, b, rbs, cues, dues, seed=17)
    sim.handle.set_obs_mode(native.OBS_TABLE if cues + dues > 512 else native.OBS_LINEAR)
    sim.step_arrays(raw)
    assert sim.check_flags() == 0
    ref = _oracle_batch(sim, pos, raw)
    for f, buf in (('sinr_db', native.BUF_SINR_DB), ('snr_db', native.BUF_SNR_DB), ('rate_bps', native.BUF_RATE_BPS),
                   ('capacity_mbps', native.BUF_CAPACITY)):
        assert rel_err(sim.fetch(buf), ref[f]) <= TOL, (shape, f)
    assert rel_err(sim.fetch(native.BUF_REWARD)[:, 0], ref['reward']) <= TOL
    assert rel_err(sim.fetch(native.BUF_OBS_TABLE), ref['table']) <= TOL
    sim.handle.close()


def test_invalid_arguments_are_rejected(native):
    """C-ABI argument checking: sizes, ranges, call order - errors, never crashes."""
    from gym_d2d_amd.simulator import Simulator
    with pytest.raises(native.NativeError, match='max_links'):
        native.Handle(num_envs=1, num_rbs=4, num_cues=2000, num_due_pairs=2000, pwr_levels_due=21, pwr_levels_cue=24,
                      pwr_levels_mbs=47)
    with pytest.raises(native.NativeError):
        native.Handle(num_envs=0, num_rbs=4, num_cues=2, num_due_pairs=2, pwr_levels_due=21, pwr_levels_cue=24,
                      pwr_levels_mbs=47)
    h = native.Handle(num_envs=2, num_rbs=4, num_cues=2, num_due_pairs=2, pwr_levels_due=21, pwr_levels_cue=24,
                      pwr_levels_mbs=47)
    with pytest.raises(native.NativeError, match='no actions'):
        h.step()                                                        # nothing configured yet
    h.upload(native.BUF_ACTIONS, np.zeros((2, 4), dtype=np.int32))
    with pytest.raises(native.NativeError, match='d2d_set_links'):
        h.step()
    with pytest.raises(native.NativeError, match='device index'):
        h.set_links([1, 99], [0, 0], [1, 1])
    with pytest.raises(native.NativeError, match='link_type'):
        h.set_links([1], [0], [7])
    with pytest.raises(native.NativeError, match='n_dev'):
        h.set_device_table(*[np.zeros(3)] * 5)
    h.set_links([1, 3], [0, 4], [1, 3])
    with pytest.raises(native.NativeError, match='positions'):
        h.upload(native.BUF_ACTIONS, np.zeros((2, 2), dtype=np.int32)); h.step()
    with pytest.raises(ValueError):
        h.set_positions(np.zeros((2, 3)), np.zeros((2, 3)))            # D is 7
    h.close()
    sim = Simulator({'num_cues': 2, 'num_due_pairs': 2, 'num_envs': 3})
    with pytest.raises(ValueError, match='single-env'):
        from gym_d2d_amd.actions import Actions
        sim.step(Actions())
    sim.set_links(sim.default_link_keys())
    with pytest.raises(ValueError, match=r'\[3,4\]'):
        sim.step_arrays(np.zeros((3, 5), dtype=np.int32))
    sim.handle.close()


def test_bucketed_and_all_pairs_paths_are_bit_identical(native):
    """Same ascending-index fmaf chain in both interference loops -> identical bits, for every reward function."""
    sim, pos, raw = _batch(native, 32, 16, 40, 60, seed=5)
    out = {}
    for bucketing in (True, False):
        sim.handle.set_bucketing(bucketing)
        for rid, param in ((native.REWARD_SYSTEM_CAPACITY, 0.5), (native.REWARD_CUE_SINR_SHANNON, 3.0)):
            sim.handle.set_reward(rid, param)
            sim.step_arrays(raw)
            out[(bucketing, rid)] = [sim.fetch(b).copy() for b in (native.BUF_SINR_DB, native.BUF_CAPACITY,
                                                                   native.BUF_REWARD, native.BUF_OBS_TABLE)]
    for rid in (native.REWARD_SYSTEM_CAPACITY, native.REWARD_CUE_SINR_SHANNON):
        for a, b in zip(out[(True, rid)], out[(False, rid)]):
            assert (a.view(np.uint32) == b.view(np.uint32)).all()
    sim.handle.close()


def test_rb_out_of_range_is_accepted_like_the_reference(native):
    """d2d_env.py:94-96 never range-checks rb; equality is all that matters.  The env is flagged and served by
    the all-pairs path."""
    sim, pos, raw = _batch(native, 4, 5, 6, 6, seed=11)
    big = raw.copy()
    big[:, ::2] += 5 * 24 * 1000         # rb far beyond num_rbs for half of the links
    sim.step_arrays(big)
    flags = sim.check_flags()
    assert flags & native.FLAG_RB_OUT_OF_RANGE and not flags & native.FLAG_ZERO_DISTANCE
    ref = _oracle_batch(sim, pos, big)
    assert rel_err(sim.fetch(native.BUF_SINR_DB), ref['sinr_db']) <= TOL
    assert (sim.fetch(native.BUF_RB) == ref['rb']).all()
    neg = raw.copy(); neg[:, 0] = -7      # Python floor semantics for negative ints
    sim.step_arrays(neg)
    ref = _oracle_batch(sim, pos, neg)
    assert (sim.fetch(native.BUF_RB) == ref['rb']).all() and (sim.fetch(native.BUF_PWR) == ref['pwr']).all()
    assert rel_err(sim.fetch(native.BUF_SINR_DB), ref['sinr_db']) <= TOL
    sim.handle.close()


def test_zero_distance_raises_value_error(native):
    """An uplink and a downlink on one RB put the BS at distance 0 from itself: the reference raises
    ValueError('math domain error') (path_loss.py:66); so does the drop-in."""
    from gym_d2d_amd.envs import D2DEnv
    env = D2DEnv({'num_rbs': 3, 'num_cues': 4, 'num_due_pairs': 2})
    env.reset()
    p_c, p_m = env.num_pwr_actions['cue'], env.num_pwr_actions['mbs']
    with pytest.raises(ValueError, match='math domain error'):
        env.step({'cue00:mbs': 1 * p_c + 3, 'mbs:cue01': 1 * p_m + 5})
    obs, *_ = env.step({'cue00:mbs': 1 * p_c + 3, 'mbs:cue01': 2 * p_m + 5})      # different RBs: fine
    assert set(obs) == {'cue00:mbs', 'mbs:cue01'} and obs['cue00:mbs'].shape == (12,)
    with pytest.raises(ZeroDivisionError):
        env.step({})                                                               # reward_fn.py:42
    with pytest.raises(TypeError):
        env.step({'cue00': 3})                                                     # key must be 'tx:rx' (d2d_env.py:76)
    with pytest.raises(KeyError):
        env.step({'cue00:nobody': 3})                                              # devices.py:28
    with pytest.raises(ValueError, match='Unable to decode'):
        env.step({'cue00:mbs': 3.5})                                               # d2d_env.py:100
    env.simulator.handle.close()


def test_hata_and_ple_power_law_accuracy(native):
    """PL_POWER path (per-tx exponent) against the oracle on a large random batch."""
    from gym_d2d_amd.path_loss import AreaType, CostHataPathLoss, LogDistancePathLoss

    class Urban(CostHataPathLoss):
        def __init__(self, f):
            super().__init__(f, AreaType.URBAN)

    class Ple(LogDistancePathLoss):
        def __init__(self, f):
            super().__init__(f, ple=3.7)

    for cls, spec in ((Urban, orc.PathLossSpec('cost_hata', 2.1, area='urban')),
                      (Ple, orc.PathLossSpec('log_distance', 2.1, ple=3.7))):
        sim, pos, raw = _batch(native, 32, 8, 20, 44, seed=3, path_loss_model=cls)
        sim.step_arrays(raw)
        ids, cfgs, is_bs = orc.device_configs(20, 44)
        cols = orc.device_columns(cfgs, is_bs)
        tx, rx, ty = default_links(20, 44)
        ref = orc.full_step(pos.astype(np.float64), tx, rx, ty, raw, cols, spec, with_obs=False)
        for f, buf in (('sinr_db', native.BUF_SINR_DB), ('snr_db', native.BUF_SNR_DB),
                       ('capacity_mbps', native.BUF_CAPACITY)):
            assert rel_err(sim.fetch(buf), ref[f]) <= TOL, (cls.__name__, f)
        sim.handle.close()


def test_shadowing_path_loss(native):
    """Native ShadowingPathLoss (PL_SHADOW): (a) same Philox draws as the oracle -> value parity; (b) per-link
    moments over many steps against the distribution captured from the reference (case13); (c) chi = 0 and links
    inside d0 are deterministic."""
    from gym_d2d_amd.path_loss import ShadowingPathLoss
    from gym_d2d_amd.simulator import Simulator
    from test_oracle_golden import check_shadow_statistics, shadow_statistics
    case = load_case('case13_shadowing')
    s = case.steps[0]
    pl = case.meta['path_loss']
    m = case.meta
    base = dict(num_rbs=m['num_rbs'], num_cues=m['num_cues'], num_due_pairs=m['num_due_pairs'], seed=4242)
    sim = Simulator(dict(base, path_loss_model=ShadowingPathLoss))
    sim.set_positions(case.pos[None].astype(np.float32))
    sim.set_links([tuple(k.split(':')) for k in s.keys])
    cols = orc.device_columns(case.cfgs, case.is_bs)
    spec = orc.PathLossSpec('log_distance', m['carrier_freq_GHz'], ple=pl['ple'])
    reps = 1500
    sinr = np.empty((reps, len(s.keys))); snr = np.empty_like(sinr)
    worst = 0.0
    for k in range(reps):
        sim.step_arrays(s.raw[None])
        sinr[k] = sim.fetch(native.BUF_SINR_DB)[0]; snr[k] = sim.fetch(native.BUF_SNR_DB)[0]
        if k < 25:
            ref = orc.step(case.pos[None], s.link_tx, s.link_rx, s.rb[None], s.pwr[None], cols, spec,
                           shadow=orc.ShadowSpec(pl['d0_m'], pl['chi_dB'], seed=4242, step=k))
            worst = max(worst, rel_err(sinr[k], ref['sinr_db'][0]), rel_err(snr[k], ref['snr_db'][0]),
                        rel_err(sim.fetch(native.BUF_CAPACITY)[0], ref['capacity_mbps'][0]))
    assert worst <= TOL, worst            # the same Philox words through the same Box-Muller: the common bar (round 6)
    check_shadow_statistics(shadow_statistics((sinr, snr)), s, reps, slack=1e-3)     # fp32 outputs
    d = np.hypot(*(case.pos[s.link_tx] - case.pos[s.link_rx]).T)
    assert (snr.std(0)[d <= pl['d0_m']] == 0).all() and (snr.std(0)[d > pl['d0_m']] > 2.0).all()
    sim.handle.close()

    class NoShadow(ShadowingPathLoss):
        def __init__(self, f):
            super().__init__(f, chi_dB=0.0)
    sim = Simulator(dict(base, path_loss_model=NoShadow))
    sim.set_positions(case.pos[None].astype(np.float32))
    sim.set_links([tuple(k.split(':')) for k in s.keys])
    sim.step_arrays(s.raw[None])
    det = orc.step(case.pos[None], s.link_tx, s.link_rx, s.rb[None], s.pwr[None], cols, spec)
    assert rel_err(sim.fetch(native.BUF_SINR_DB)[0], det['sinr_db'][0]) <= TOL
    sim.handle.close()


def test_custom_python_path_loss_with_a_batch(native):
    """The Python-plugin route for B > 1: the user's PathLoss is evaluated per env for the (transmitter of a link) x (receiver of
    a link) pairs into a [B, N, N] table by link pair (d2d_set_path_loss_link_table, per_env = 1) after every position change."""
    import math
    from gym_d2d_amd.path_loss import PathLoss
    from gym_d2d_amd.simulator import Simulator

    class TwoSlope(PathLoss):                      # not a single power law: cannot be lowered, must go through the table
        def __call__(self, tx, rx):
            d = tx.position.distance(rx.position)
            base = 40.0 + 20.0 * math.log10(d)
            return base if d < 50.0 else base + 15.0 * math.log10(d / 50.0) + 0.5 * (tx.antenna_height_m - rx.antenna_height_m)

    b, cues, dues, rbs = 3, 4, 5, 3
    rng = np.random.default_rng(8)
    sim = Simulator(dict(num_rbs=rbs, num_cues=cues, num_due_pairs=dues, num_envs=b, path_loss_model=TwoSlope))
    pos = random_layout(rng, b, cues, dues)
    sim.set_positions(pos)
    sim.set_links(sim.default_link_keys())
    raw = np.concatenate([rng.integers(0, rbs * 24, (b, cues)), rng.integers(0, rbs * 21, (b, dues))], 1).astype(np.int32)
    sim.step_arrays(raw)
    assert sim.check_flags() & native.FLAG_ZERO_DISTANCE == 0
    ids, cfgs, is_bs = orc.device_configs(cues, dues)
    cols = orc.device_columns(cfgs, is_bs)
    p64 = pos.astype(np.float64)
    d = np.hypot(p64[:, :, None, 0] - p64[:, None, :, 0], p64[:, :, None, 1] - p64[:, None, :, 1])
    with np.errstate(divide='ignore'):
        base = 40.0 + 20.0 * np.log10(d)
        table = np.where(d < 50.0, base, base + 15.0 * np.log10(d / 50.0) +
                         0.5 * (cols.ant_h_m[None, :, None] - cols.ant_h_m[None, None, :]))
    tx, rx, ty = default_links(cues, dues)
    ref = orc.full_step(p64, tx, rx, ty, raw, cols, orc.PathLossSpec('table', 2.1, table_db=table), with_obs=False)
    for f, buf in (('sinr_db', native.BUF_SINR_DB), ('snr_db', native.BUF_SNR_DB), ('capacity_mbps', native.BUF_CAPACITY)):
        assert rel_err(sim.fetch(buf), ref[f]) <= 1e-5, f      # the table travels as float64 dB (ABI 3): the usual bar
    sim.handle.close()


def test_device_reset_matches_oracle_sampler(native):
    """csrc/d2d_reset.hip against the oracle's sampler fed with the same Philox uniforms + the reference's own
    sampler properties (test_position.py:30-44)."""
    from gym_d2d_amd.simulator import Simulator
    b, cues, dues = 256, 7, 9
    sim = Simulator(dict(num_cues=cues, num_due_pairs=dues, num_envs=b, cell_radius_m=500.0, d2d_radius_m=20.0))
    sim.handle.set_env_offset(1000)
    sim.reset_device(seed=0x1234_5678_9ABC, episode=3)
    got = sim.positions().astype(np.float64)
    d = 1 + cues + 2 * dues
    u = orc.reset_uniforms(0x1234_5678_9ABC, 3, b, d, 64, first_env=1000)
    ref, used = orc.sample_positions_from_uniforms(u, cues, dues, 500.0, 20.0)
    # fp32 sincos/sqrt vs fp64: 1e-6 of the cell radius; a rejection decision may differ only on the boundary
    close = np.abs(got - ref).max(axis=2) <= 500.0 * 1e-6
    r_ref = np.hypot(ref[..., 0], ref[..., 1])
    assert (close | (np.abs(r_ref - 500.0) < 1e-3)).all()
    assert close.mean() > 0.999
    assert (got[:, 0] == 0).all()
    assert (np.hypot(got[..., 0], got[..., 1]) <= 500.0 * (1 + 1e-6)).all()
    tx = got[:, 1 + cues::2]; rx = got[:, 2 + cues::2]
    assert (np.hypot(tx[..., 0] - rx[..., 0], tx[..., 1] - rx[..., 1]) <= 20.0 * (1 + 1e-5)).all()
    assert used.max() > 1, 'rejection loop should be exercised'
    # different episode -> different layout; same (seed, episode) -> identical bits
    sim.reset_device(seed=0x1234_5678_9ABC, episode=4)
    assert (sim.positions() != got.astype(np.float32)).any()
    sim.reset_device(seed=0x1234_5678_9ABC, episode=3)
    assert (sim.positions() == got.astype(np.float32)).all()
    sim.handle.close()


def test_vec_env_episode(native):
    """VecD2DEnv end to end (torch-bound buffers when torch sees the GPU): reset + 10 steps against the oracle."""
    import torch
    from gym_d2d_amd.envs import VecD2DEnv
    env = VecD2DEnv({'num_rbs': 6, 'num_cues': 5, 'num_due_pairs': 8, 'seed': 77}, num_envs=48)
    obs = env.reset()
    assert tuple(obs.shape) == (48, 13, 78)
    pos = env.simulator.positions()
    rng = np.random.default_rng(0)
    tx, rx, ty = default_links(5, 8)
    ids, cfgs, is_bs = orc.device_configs(5, 8)
    cols = orc.device_columns(cfgs, is_bs)
    for k in range(10):
        raw = np.concatenate([rng.integers(0, 6 * 24, (48, 5)), rng.integers(0, 6 * 21, (48, 8))], 1).astype(np.int32)
        act = torch.as_tensor(raw, device=obs.device) if torch.is_tensor(obs) else raw
        obs, rew, dones, info = env.step(act)
        ref = orc.full_step(pos.astype(np.float64), tx, rx, ty, raw, cols, orc.PathLossSpec())
        to_np = lambda t: t.cpu().numpy() if torch.is_tensor(t) else np.asarray(t)
        assert rel_err(to_np(obs), ref['obs']) <= TOL
        assert rel_err(to_np(rew), np.repeat(ref['reward'][:, None], 13, 1)) <= TOL
        assert rel_err(to_np(info['sinr_db']), ref['sinr_db']) <= TOL
        assert bool(to_np(dones).all()) == (k == 9)
    assert env.status_flags() == 0
    env.close()


def test_vec_env_traffic_mode_matches_reference_traffic_model(native):
    """VecD2DEnv(cue_actions='traffic'): CUE rb/pwr from UplinkTrafficModel, agents act for DUEs only - against the
    reference's own UplinkTrafficModel.get_traffic fed through its Simulator.step (golden case14)."""
    from gym_d2d_amd.envs import VecD2DEnv
    case = load_case('case14_traffic_model')
    m = case.meta
    env = VecD2DEnv({'num_rbs': m['num_rbs'], 'num_cues': m['num_cues'], 'num_due_pairs': m['num_due_pairs']},
                    num_envs=1, cue_actions='traffic', use_torch=False)
    env.reset(seed=0)
    env.simulator.set_positions(case.pos[None].astype(np.float32))
    for s in case.steps:
        obs, rew, dones, info = env.step(s.due_raw[None].astype(np.int32))
        assert [f'{t}:{r}' for t, r in env.simulator.link_keys] == s.keys
        assert (info['rb'][0] == s.rb).all() and (info['tx_pwr_dbm'][0] == s.pwr).all()
        for f in ('sinr_db', 'snr_db', 'rate_bps', 'capacity_mbps'):
            assert rel_err(info[f][0], getattr(s, f)) <= TOL, f
        assert rel_err(rew[0], s.reward_system_capacity) <= TOL
        assert rel_err(obs[0][s.obs_rows], s.obs) <= TOL
    env.close()


def test_vec_env_downlink_traffic_model(native):
    """cue_actions='traffic' with DownlinkTrafficModel (traffic_model.py:25-32): links mbs -> cueXX, rb = i mod R,
    power = the CUE's max power, decoded with the base station's 47-level alphabet; against the oracle."""
    from gym_d2d_amd.envs import VecD2DEnv
    from gym_d2d_amd.traffic_model import DownlinkTrafficModel
    env = VecD2DEnv({'num_rbs': 5, 'num_cues': 7, 'num_due_pairs': 6, 'traffic_model': DownlinkTrafficModel},
                    num_envs=32, cue_actions='traffic', use_torch=False)
    env.reset(seed=4)
    pos = env.simulator.positions()
    due = np.random.default_rng(2).integers(0, 5 * 21, (32, 6)).astype(np.int32)
    obs, rew, dones, info = env.step(due)
    assert env.status_flags() == 0                      # downlinks only: the BS never receives
    assert (info['rb'][:, :7] == np.arange(7) % 5).all() and (info['tx_pwr_dbm'][:, :7] == 23).all()
    tx = np.array([0] * 7 + [8 + 2 * p for p in range(6)]); rx = np.array(list(range(1, 8)) + [9 + 2 * p for p in range(6)])
    assert (tx == env.simulator.link_tx).all() and (rx == env.simulator.link_rx).all()
    ids, cfgs, is_bs = orc.device_configs(7, 6)
    st = orc.step(pos.astype(np.float64), tx, rx, info['rb'], info['tx_pwr_dbm'], orc.device_columns(cfgs, is_bs),
                  orc.PathLossSpec())
    for f in ('sinr_db', 'snr_db', 'capacity_mbps'):
        assert rel_err(info[f], st[f]) <= TOL, f
    ty = np.array([2] * 7 + [3] * 6)
    assert rel_err(rew[:, 0], orc.reward_system_capacity(st['capacity_mbps'], info['rb'], ty)) <= TOL
    env.close()


def test_vec_env_is_stream_ordered_with_torch(native):
    """The library's kernels run on torch's current stream: actions produced by (slow) torch work queued just before
    step() must be the ones decoded, and results must be visible to torch ops queued right after - on the default
    stream and on a side stream."""
    import torch
    from gym_d2d_amd.envs import VecD2DEnv
    env = VecD2DEnv({'num_rbs': 7, 'num_cues': 9, 'num_due_pairs': 11}, num_envs=512)
    obs = env.reset(seed=3)
    dev = obs.device
    for stream in (torch.cuda.current_stream(dev), torch.cuda.Stream(device=dev)):
        with torch.cuda.stream(stream):
            for k in range(3):
                torch.cuda._sleep(200_000_000)                          # ~0.1 s of GPU time ahead of the producer
                act = torch.randint(0, 7 * 21, (512, 20), device=dev, dtype=torch.int32)
                obs, rew, dones, info = env.step(act)
                got_rb = info['rb'].clone()                             # consumer queued right behind the step
                want_cue = act[:, :9] // 24
                want_due = act[:, 9:] // 21
                assert torch.equal(got_rb[:, :9], want_cue) and torch.equal(got_rb[:, 9:], want_due), (str(stream), k)
        stream.synchronize()
    env.close()


def test_step_is_hip_graph_capturable(native):
    """Steady-state d2d_step allocates nothing and never synchronises, so a caller may capture a rollout into a HIP
    graph (torch.cuda.graph) and replay it; the replay reproduces the eager results bit for bit."""
    import torch
    from gym_d2d_amd.envs import VecD2DEnv
    env = VecD2DEnv({'num_rbs': 9, 'num_cues': 10, 'num_due_pairs': 14}, num_envs=256)
    env.reset(seed=2)
    h, dev = env.simulator.handle, env.device
    acts = torch.randint(0, 9 * 21, (5, 256, 24), device=dev, dtype=torch.int32)
    for k in range(5):
        h.step(acts[k].data_ptr())
    torch.cuda.synchronize()
    want = {k: env._t[k].clone() for k in ('sinr_db', 'reward', 'obs')}
    side = torch.cuda.Stream(device=dev)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        env._follow_torch_stream()
        h.step(acts[0].data_ptr())                      # buffers and tables are in place before capture
        side.synchronize()
        with torch.cuda.graph(graph, stream=side):
            for k in range(5):
                h.step(acts[k].data_ptr())
    torch.cuda.synchronize()
    for t in want:
        env._t[t].zero_()
    graph.replay()
    torch.cuda.synchronize()
    for t, ref in want.items():
        assert torch.equal(env._t[t], ref), t
    env._follow_torch_stream()
    env.close()


def test_vec_env_plugin_swap(native):
    """BASELINE config 4: FreeSpacePathLoss through the plugin route + a custom array ObsFunction + Shannon reward."""
    import torch
    from gym_d2d_amd.envs import VecD2DEnv
    from gym_d2d_amd.envs.obs_fn import OwnLinkObsFunction
    from gym_d2d_amd.envs.reward_fn import ShannonRewardFunction
    from gym_d2d_amd.path_loss import FreeSpacePathLoss
    cfg = {'num_rbs': 8, 'num_cues': 6, 'num_due_pairs': 10, 'path_loss_model': FreeSpacePathLoss,
           'obs_fn': OwnLinkObsFunction, 'reward_fn': ShannonRewardFunction}
    env = VecD2DEnv(cfg, num_envs=16, cue_actions='traffic')
    obs = env.reset(seed=5)
    assert tuple(obs.shape) == (16, 16, 6)
    rng = np.random.default_rng(1)
    due = rng.integers(0, 8 * 21, (16, 10)).astype(np.int32)
    obs, rew, dones, info = env.step(torch.as_tensor(due, device=obs.device) if torch.is_tensor(obs) else due)
    to_np = lambda t: t.cpu().numpy() if torch.is_tensor(t) else np.asarray(t)
    # CUE links follow UplinkTrafficModel: rb = i mod R at max power 23 dBm (traffic_model.py:15-22)
    assert (to_np(info['rb'])[:, :6] == np.arange(6) % 8).all() and (to_np(info['tx_pwr_dbm'])[:, :6] == 23).all()
    pos = env.simulator.positions()
    tx, rx, ty = default_links(6, 10)
    ids, cfgs, is_bs = orc.device_configs(6, 10)
    cols = orc.device_columns(cfgs, is_bs)
    st = orc.step(pos.astype(np.float64), tx, rx, to_np(info['rb']), to_np(info['tx_pwr_dbm']), cols, orc.PathLossSpec())
    assert rel_err(to_np(info['sinr_db']), st['sinr_db']) <= TOL
    assert rel_err(to_np(rew), orc.reward_shannon(st['sinr_db'])) <= TOL
    assert rel_err(to_np(obs), orc.obs_table(pos.astype(np.float64), tx, rx, st['sinr_db'], st['snr_db'])) <= TOL
    env.close()
