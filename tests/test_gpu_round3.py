"""GPU tests of the round-3 additions, through the C ABI: the per-RB member-list interferer search (D2D_TUNE_STEP_WALK = 2)
is bit-identical to the mask walk and the all-pairs sweep for every path-loss mode, reward and launch geometry - list
overflow (more than eight links on one RB) included; d2d_set_export_actions; obs_dtype."""
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent

from golden_util import rel_err
from oracle import d2d_oracle as orc
from sim_util import default_links, random_layout

pytestmark = pytest.mark.gpu
TOL = 1e-5

OUTS = ('BUF_SINR_DB', 'BUF_SNR_DB', 'BUF_RATE_BPS', 'BUF_CAPACITY', 'BUF_REWARD', 'BUF_OBS_TABLE', 'BUF_RB', 'BUF_PWR',
        'BUF_ENV_FLAGS')


@pytest.fixture(scope='module')
def native():
    from gym_d2d_amd import _native
    _native.load_library()
    return _native


def _batch(num_envs, rbs, cues, dues, rng_seed, **cfg):
    from gym_d2d_amd.simulator import Simulator
    rng = np.random.default_rng(rng_seed)
    sim = Simulator(dict(num_rbs=rbs, num_cues=cues, num_due_pairs=dues, num_envs=num_envs, **cfg))
    pos = random_layout(rng, num_envs, cues, dues)
    sim.set_positions(pos)
    sim.set_links(sim.default_link_keys())
    p = sim.config.num_pwr_actions
    raw = np.concatenate([rng.integers(0, rbs * p['cue'], (num_envs, cues)),
                          rng.integers(0, rbs * p['due'], (num_envs, dues))], axis=1).astype(np.int32)
    return sim, pos, raw


def _snapshot(sim, native, with_obs=False):
    out = {name: sim.fetch(getattr(native, name)).copy() for name in OUTS}
    if with_obs:
        out['BUF_OBS'] = sim.fetch(native.BUF_OBS).copy()
    return out


def _variants(native, h, fn):
    """fn() once per interferer-search variant; returns {name: snapshot}."""
    out = {}
    for name, bucket, walk in (('mask_walk', True, 0), ('member_lists', True, 2), ('all_pairs', False, 0), ('auto', True, -1)):
        h.set_bucketing(bucket)
        h.set_tuning(native.TUNE_STEP_WALK, walk)
        out[name] = fn()
    h.set_bucketing(True)
    h.set_tuning(native.TUNE_STEP_WALK, -1)
    return out


def _same(snaps, ref_name='mask_walk'):
    for name, snap in snaps.items():
        for buf, ref in snaps[ref_name].items():
            assert np.array_equal(snap[buf], ref, equal_nan=True), (name, buf)


# shapes: lists never overflow / a few envs overflow / every env overflows (25, 18 links per RB) / N > 512 (two links per
# thread) / one link per RB / a single RB
SHAPES = [(33, 25, 25, 25), (64, 256, 256, 256), (16, 4, 40, 60), (8, 7, 0, 130), (5, 300, 100, 91), (9, 1, 64, 64),
          (6, 64, 300, 400), (3, 500, 600, 600)]


@pytest.mark.parametrize('shape', SHAPES)
@pytest.mark.parametrize('reward', [0, 1, 2, 3])
def test_member_lists_are_bit_identical_to_masks_and_all_pairs(native, shape, reward):
    b, rbs, cues, dues = shape
    sim, pos, raw = _batch(b, rbs, cues, dues, rng_seed=sum(shape) + reward)
    h = sim.handle
    h.set_obs_mode(native.OBS_TABLE)
    h.set_reward(reward, {0: 0.0, 1: 0.0, 2: -70.0, 3: 0.0}[reward])

    if reward == 0:
        h.upload(native.BUF_REWARD, np.zeros((b, cues + dues), np.float32))      # not written without a reward function

    def run():
        sim.step_arrays(raw)
        return _snapshot(sim, native)
    snaps = _variants(native, h, run)
    _same(snaps)
    tx, rx, ty = default_links(cues, dues)
    ref = orc.full_step(pos.astype(np.float64), tx, rx, ty, raw, orc.device_columns(*orc.device_configs(cues, dues)[1:]),
                        orc.PathLossSpec(), with_obs=False, chunk=8)
    assert rel_err(snaps['member_lists']['BUF_SINR_DB'], ref['sinr_db']) <= TOL
    assert rel_err(snaps['member_lists']['BUF_CAPACITY'], ref['capacity_mbps']) <= TOL
    if reward == 1:
        assert rel_err(snaps['member_lists']['BUF_REWARD'][:, 0], ref['reward']) <= TOL
    sim.handle.close()


def test_member_lists_with_skewed_actions_and_out_of_range_rbs(native):
    """Action distributions a policy can produce: every DUE on one of 3 RBs (lists overflow in every env), exactly two links
    per RB (no overflow anywhere), and envs with rb outside [0, R) (those links take the sweep, the others their lists)."""
    b, rbs, cues, dues = 40, 64, 64, 64
    sim, pos, raw = _batch(b, rbs, cues, dues, rng_seed=77)
    h = sim.handle
    h.set_obs_mode(native.OBS_TABLE)
    h.set_reward(1, 0.0)
    p = sim.config.num_pwr_actions
    rng = np.random.default_rng(3)
    raw[:10, cues:] = rng.integers(0, 3, (10, dues)) * p['due'] + rng.integers(0, p['due'], (10, dues))
    raw[10:20, :cues] = np.arange(cues)[None] * p['cue'] + 5
    raw[10:20, cues:] = np.arange(dues)[None] * p['due'] + 7
    raw[20:25, 3] = rbs * p['cue'] + 11               # rb == R: out of range
    raw[25:30, cues + 5] = -4                         # negative action: rb = -1 (Python floor)

    def run():
        sim.step_arrays(raw)
        return _snapshot(sim, native)
    snaps = _variants(native, h, run)
    _same(snaps)
    flags = snaps['member_lists']['BUF_ENV_FLAGS']
    assert (flags[20:30] & native.FLAG_RB_OUT_OF_RANGE).all() and not (flags[:20] & native.FLAG_RB_OUT_OF_RANGE).any()
    sim.handle.close()


@pytest.mark.parametrize('model', ['ple35', 'cost_hata', 'table', 'shadowing'])
def test_member_lists_every_path_loss_mode(native, model):
    from gym_d2d_amd import path_loss as pl

    class Ple35(pl.LogDistancePathLoss):
        def __init__(self, carrier_freq_GHz):
            super().__init__(carrier_freq_GHz, 3.5)

    class Plugin(pl.PathLoss):                       # evaluated on the host -> [D, D] table route
        def __call__(self, tx, rx):
            return 30.0 + 31.0 * np.log10(tx.position.distance(rx.position)) - 0.5 * tx.tx_antenna_gain_dBi

    cls = {'ple35': Ple35, 'cost_hata': pl.CostHataPathLoss, 'table': Plugin, 'shadowing': pl.ShadowingPathLoss}[model]
    b, rbs, cues, dues = (12, 6, 20, 30) if model != 'table' else (1, 6, 20, 30)
    sim, pos, raw = _batch(b, rbs, cues, dues, rng_seed=5, path_loss_model=cls, **({'seed': 99} if model == 'shadowing' else {}))
    h = sim.handle
    h.set_obs_mode(native.OBS_TABLE)
    for reward in (1, 2):
        h.set_reward(reward, {1: 0.0, 2: -70.0}[reward])

        def run():
            if model == 'shadowing':                # the draws are keyed by (seed, step): restart the stream per variant
                sim._install_tables()
            sim.step_arrays(raw)
            return _snapshot(sim, native)
        _same(_variants(native, h, run))
    sim.handle.close()


@pytest.mark.parametrize('shape', [(13, 25, 25, 25), (7, 5, 9, 10), (10, 3, 30, 37)])
def test_member_lists_small_envs_sharing_a_workgroup_and_fused_obs(native, shape):
    """Several envs per workgroup: an overflow in ONE env sends the whole workgroup through the mask fallback; the fused
    LinearObs expansion rides on either path.  Traffic-model CUEs as in BASELINE config 2."""
    b, rbs, cues, dues = shape
    sim, pos, raw = _batch(b, rbs, cues, dues, rng_seed=sum(shape))
    h = sim.handle
    h.set_obs_mode(native.OBS_LINEAR)
    ref = None
    for walk in (0, 2):
        h.set_tuning(native.TUNE_STEP_WALK, walk)
        for epw, fuse in ((1, 0), (2, 1), (4, 1), (0, -1)):
            h.set_tuning(native.TUNE_STEP_ENVS_PER_WG, epw)
            h.set_tuning(native.TUNE_STEP_FUSE_OBS, fuse)
            h.upload(native.BUF_OBS, np.full((b, cues + dues, 6 * (cues + dues)), np.nan, np.float32))
            sim.step_arrays(raw)
            snap = _snapshot(sim, native, True)
            ref = ref or snap
            for buf, r in ref.items():
                assert np.array_equal(snap[buf], r, equal_nan=True), (walk, epw, fuse, buf)
    sim.handle.close()


def test_baseline_config_2_rollout_kernel_with_member_lists(native):
    """The level-2 rollout specialisation (traffic-model prefix, fused 16-byte expansion) on lists vs masks, 1024 x 50."""
    import torch
    from gym_d2d_amd.envs import VecD2DEnv
    b, c, p, r = 1024, 25, 25, 25
    snaps = {}
    rng = np.random.default_rng(5)
    acts = rng.integers(0, r * 21, (3, b, p)).astype(np.int32)
    for walk in (0, 2):
        env = VecD2DEnv({'num_rbs': r, 'num_cues': c, 'num_due_pairs': p}, num_envs=b, cue_actions='traffic')
        env.simulator.handle.set_tuning(native.TUNE_STEP_WALK, walk)
        env.reset(seed=2024)
        outs = []
        for k in range(3):
            obs, rew, _, info = env.step(torch.as_tensor(acts[k], device=env.device))
            outs.append([obs.cpu().numpy().copy(), rew.cpu().numpy().copy(), info['sinr_db'].cpu().numpy().copy(),
                         info['rb'].cpu().numpy().copy()])
        snaps[walk] = outs
        env.close()
    for k in range(3):
        for x, y in zip(snaps[0][k], snaps[2][k]):
            assert np.array_equal(x, y)


def test_full_size_member_lists_against_masks(native):
    """4096 x 512 (BASELINE config 3), the rollout specialisation: lists vs masks, every output of every env; about 6 % of
    the envs overflow a list under uniformly random actions."""
    import torch
    from gym_d2d_amd.envs import VecD2DEnv
    from gym_d2d_amd.envs.obs_fn import OwnLinkObsFunction
    b, c, p, r = 4096, 256, 256, 256
    env = VecD2DEnv({'num_rbs': r, 'num_cues': c, 'num_due_pairs': p, 'obs_fn': OwnLinkObsFunction}, num_envs=b)
    env.reset(seed=11)
    h = env.simulator.handle
    g = torch.Generator(device=env.device); g.manual_seed(4)
    act = torch.randint(0, r * 21, (b, c + p), generator=g, device=env.device, dtype=torch.int32)
    names = ('sinr_db', 'snr_db', 'rate_bps', 'capacity_mbps', 'reward', 'table', 'rb', 'pwr', 'env_flags')
    got = {}
    for walk in (0, 2):
        h.set_tuning(native.TUNE_STEP_WALK, walk)
        env.step(act)
        torch.cuda.synchronize()
        got[walk] = {n: env._t[n].clone() for n in names}
    for n in names:
        assert torch.equal(got[0][n], got[2][n]), n
    # how many envs overflowed: count RBs with more than 8 links
    rb = got[2]['rb'].long()
    counts = torch.zeros((b, r), dtype=torch.long, device=env.device).scatter_add_(1, rb, torch.ones_like(rb))
    over = int((counts.max(dim=1).values > 8).sum())
    assert 0 < over < b // 4, over
    env.close()


def test_export_actions_switch(native):
    """d2d_set_export_actions(0): D2D_BUF_RB / PWR keep their last contents, every other output is unchanged."""
    sim, pos, raw = _batch(16, 32, 48, 48, rng_seed=9)
    h = sim.handle
    h.set_obs_mode(native.OBS_TABLE)
    sim.step_arrays(raw)
    ref = _snapshot(sim, native)
    marker = np.full_like(ref['BUF_RB'], -123)
    for walk in (0, 2):
        h.set_tuning(native.TUNE_STEP_WALK, walk)
        h.upload(native.BUF_RB, marker); h.upload(native.BUF_PWR, marker)
        h.set_export_actions(False)
        sim.step_arrays(raw)
        snap = _snapshot(sim, native)
        assert (snap['BUF_RB'] == -123).all() and (snap['BUF_PWR'] == -123).all()
        for buf in OUTS:
            if buf not in ('BUF_RB', 'BUF_PWR'):
                assert np.array_equal(snap[buf], ref[buf]), buf
        h.set_export_actions(True)
        sim.step_arrays(raw)
        snap = _snapshot(sim, native)
        for buf in OUTS:
            assert np.array_equal(snap[buf], ref[buf]), buf
    sim.handle.close()


def test_rollout_kernel_options_are_bit_identical(native):
    """The rollout specialisation's options - link records by scalar loads (legal when every aligned group of 64 links has
    identical records), nontemporal result stores - change no bit of any output; with a per-device override the records
    stop being uniform, the library drops the scalar loads by itself, and the results still equal the generic kernel's."""
    import json
    import tempfile
    from pathlib import Path
    from gym_d2d_amd.simulator import Simulator
    b, rbs, cues, dues = 48, 128, 128, 128            # N = 256 = one link per thread, one env per workgroup
    sim, pos, raw = _batch(b, rbs, cues, dues, rng_seed=21)
    h = sim.handle
    h.set_obs_mode(native.OBS_TABLE)
    ref = None
    for srec in (0, 1):
        for nt in (0, 1):
            h.set_tuning(native.TUNE_STEP_SCALAR_RECORDS, srec)
            h.set_tuning(native.TUNE_STEP_NT_RESULTS, nt)
            sim.step_arrays(raw)
            snap = _snapshot(sim, native)
            ref = ref or snap
            for buf, r in ref.items():
                assert np.array_equal(snap[buf], r), (srec, nt, buf)
    h.set_bucketing(False)                             # generic kernel, all-pairs sweep
    sim.step_arrays(raw)
    for buf, r in _snapshot(sim, native).items():
        assert np.array_equal(ref[buf], r), buf
    tx, rx, ty = default_links(cues, dues)
    want = orc.full_step(pos.astype(np.float64), tx, rx, ty, raw, orc.device_columns(*orc.device_configs(cues, dues)[1:]),
                         orc.PathLossSpec(), with_obs=False, chunk=8)
    assert rel_err(ref['BUF_SINR_DB'], want['sinr_db']) <= TOL and rel_err(ref['BUF_REWARD'][:, 0], want['reward']) <= TOL
    sim.handle.close()
    # one CUE with its own antenna gain: records no longer uniform within its group of 64
    with tempfile.TemporaryDirectory() as tmp:
        path = Path(tmp) / 'devices.json'
        path.write_text(json.dumps({'cue07': {'config': {'tx_antenna_gain_dBi': 3.5}}}))
        rng = np.random.default_rng(21)
        sim = Simulator(dict(num_rbs=rbs, num_cues=cues, num_due_pairs=dues, num_envs=b, device_config_file=path))
        sim.set_positions(pos)
        sim.set_links(sim.default_link_keys())
        h = sim.handle
        h.set_obs_mode(native.OBS_TABLE)
        outs = []
        for srec, bucket in ((1, True), (0, True), (0, False)):
            h.set_tuning(native.TUNE_STEP_SCALAR_RECORDS, srec)
            h.set_bucketing(bucket)
            sim.step_arrays(raw)
            outs.append(_snapshot(sim, native))
        for other in outs[1:]:
            for buf, r in outs[0].items():
                assert np.array_equal(other[buf], r), buf
        assert not np.array_equal(outs[0]['BUF_SINR_DB'], ref['BUF_SINR_DB'])       # the override reached the kernel
        sim.handle.close()


def test_gym_make_builds_a_working_env(tmp_path):
    """gym.make('D2DEnv-v0', env_config=...) through a stand-in `gym` package (tests/gym_stub_util.py): registration at
    import (gym_d2d/__init__.py:8-11), a gym.Env subclass, reset / step with the reference's dict conventions."""
    from pathlib import Path
    from gym_stub_util import run_gym_make
    out = run_gym_make(tmp_path, Path(__file__).resolve().parent.parent)
    assert out['entry_point'] == 'gym_d2d_amd.envs:D2DEnv' and out['make'] == 'ok' and out['is_gym_env'] is True
    assert out['agents'] == 7 and out['obs_width'] == 42 and out['obs_space'] == [42] and out['done'] == {'__all__': False}
    assert out['info_keys'] == ['capacity_mbps', 'rate_bps', 'rb', 'sinr_db', 'snr_db', 'tx_pwr_dbm']


def test_vec_env_obs_dtype_and_export_switch(native):
    """env_config['obs_dtype'] = 'float64' returns the reference's observation dtype (obs_fn.py:51) with the float32 values;
    export_actions=False leaves info['rb'] / info['tx_pwr_dbm'] out and every other output unchanged."""
    import torch
    from gym_d2d_amd.envs import VecD2DEnv
    cfg = {'num_rbs': 6, 'num_cues': 5, 'num_due_pairs': 9}
    a = VecD2DEnv(dict(cfg), num_envs=32)
    b64 = VecD2DEnv(dict(cfg, obs_dtype='float64'), num_envs=32, export_actions=False)
    oa, ob = a.reset(seed=3), b64.reset(seed=3)
    assert oa.dtype == torch.float32 and ob.dtype == torch.float64 and torch.equal(oa.double(), ob)
    act = torch.randint(0, 6 * 21, (32, 14), device=a.device, dtype=torch.int32)
    ra, rb_ = a.step(act), b64.step(act)
    assert rb_[0].dtype == torch.float64 and torch.equal(ra[0].double(), rb_[0]) and torch.equal(ra[1], rb_[1])
    assert rb_[3]['rb'] is None and rb_[3]['tx_pwr_dbm'] is None and ra[3]['rb'] is not None
    assert torch.equal(ra[3]['sinr_db'], rb_[3]['sinr_db']) and torch.equal(ra[2], rb_[2])
    with pytest.raises(ValueError):
        VecD2DEnv(dict(cfg, obs_dtype='float16'), num_envs=2)
    a.close(); b64.close()


def test_maximum_links_per_env_against_the_c_oracle(native):
    """D2D_MAX_LINKS = 2048 links per env (1024 CUEs + 1024 DUE pairs: two links per thread, member lists by default because
    the masks stop at 1024 links) - every link of every env against the plain-C oracle, for a spread-out RB choice (lists)
    and a crowded one (64 RBs: 32 links per RB overflow every list -> the sweep)."""
    from oracle import c_oracle
    for rbs in (1024, 64):
        b, cues, dues = 6, 1024, 1024
        sim, pos, raw = _batch(b, rbs, cues, dues, rng_seed=rbs)
        h = sim.handle
        h.set_obs_mode(native.OBS_TABLE)
        sim.step_arrays(raw)
        got = _snapshot(sim, native)
        assert (got['BUF_ENV_FLAGS'] == 0).all()
        tx, rx, ty = default_links(cues, dues)
        ref = c_oracle.full_step(pos.astype(np.float64), tx, rx, ty, raw, orc.device_columns(*orc.device_configs(cues, dues)[1:]),
                                 orc.PathLossSpec(), with_obs=False, threads=min(8, c_oracle.max_threads()))
        assert np.array_equal(got['BUF_RB'], ref['rb']) and np.array_equal(got['BUF_PWR'], ref['pwr'])
        for buf, f in (('BUF_SINR_DB', 'sinr_db'), ('BUF_SNR_DB', 'snr_db'), ('BUF_RATE_BPS', 'rate_bps'), ('BUF_CAPACITY', 'capacity_mbps')):
            assert rel_err(got[buf], ref[f]) <= TOL, (rbs, f)
        assert rel_err(got['BUF_REWARD'][:, 0], ref['reward']) <= TOL and rel_err(got['BUF_OBS_TABLE'], ref['table']) <= TOL
        h.set_bucketing(False)                             # the all-pairs sweep: the same bits
        sim.step_arrays(raw)
        for buf, r in _snapshot(sim, native).items():
            assert np.array_equal(got[buf], r), (rbs, buf)
        sim.handle.close()
    from gym_d2d_amd.simulator import Simulator
    with pytest.raises(Exception):                          # one link more than the library takes
        s = Simulator(dict(num_rbs=4, num_cues=1025, num_due_pairs=1024, num_envs=1))
        s.set_links(s.default_link_keys())


def test_write_ceiling_probe_family(native):
    """libd2d_probe.so (include/d2d_hip_diag.h; measurement equipment, not the product library): every variant of the fill
    family reports a plausible rate, the best is the maximum, the obs kernel's own geometry is variant 0."""
    import sys
    sys.path.insert(0, str(ROOT / 'tools'))
    import write_probe
    best, rates = write_probe.write_variants(1 << 30, 3)
    assert len(rates) == 33 and all(500.0 < r < 8000.0 for r in rates), rates
    assert abs(best - max(rates)) < 1e-6
    with pytest.raises(ValueError):
        write_probe.write_variants(1 << 20, 1)              # below one group of regions


@pytest.mark.parametrize('shape', [(3, 100000, 25, 25), (2, 5000, 300, 300), (1, 1, 0, 1), (1, 1, 1, 0), (2, 70000, 1000, 1000), (4, 1, 1, 1)])
def test_edge_shapes_against_the_oracle(native, shape):
    """More resource blocks than any per-RB structure fits in LDS (masks and lists both give way to the sweep), a single link, one
    RB for everything, 2000 links on 70000 RBs: SINR, reward and flags against the oracle."""
    b, rbs, cues, dues = shape
    sim, pos, raw = _batch(b, rbs, cues, dues, rng_seed=sum(shape))
    h = sim.handle
    h.set_obs_mode(native.OBS_LINEAR if cues + dues <= 128 else native.OBS_TABLE)
    sim.step_arrays(raw)
    tx, rx, ty = default_links(cues, dues)
    ref = orc.full_step(pos.astype(np.float64), tx, rx, ty, raw, orc.device_columns(*orc.device_configs(cues, dues)[1:]), orc.PathLossSpec(),
                        with_obs=False, chunk=2)
    assert rel_err(sim.fetch(native.BUF_SINR_DB), ref['sinr_db']) <= TOL
    assert rel_err(sim.fetch(native.BUF_REWARD)[:, 0], ref['reward']) <= TOL
    assert h.status_flags() == 0
    h.close()
