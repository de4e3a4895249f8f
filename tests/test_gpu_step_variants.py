"""The step kernels' variants against each other and the oracle, through the C ABI: the interferer search (membership masks,
per-RB member lists, masked all-pairs sweep), the rollout kernel (csrc/d2d_rollout.hip) and its options, one or two links per
thread, the compile-time specialisations, the action decode, edge shapes.  Every variant must produce the same BITS."""
from pathlib import Path

import numpy as np
import pytest

from golden_util import load_case, rel_err
from oracle import d2d_oracle as orc
from sim_util import OUTS, assert_same as _same, default_links, random_batch as _batch, random_layout, search_variants as _variants, snapshot as _snapshot

ROOT = Path(__file__).resolve().parent.parent
pytestmark = pytest.mark.gpu
TOL = 1e-5


@pytest.mark.parametrize('shape', [(33, 25, 25, 25), (24, 256, 256, 256), (16, 4, 40, 60), (8, 7, 0, 130), (5, 300, 100, 91),
                                   (9, 1, 64, 64)])
@pytest.mark.parametrize('reward', [1, 2, 3])
def test_interferer_search_variants_are_bit_identical(native, shape, reward):
    """The bitmask walk and the masked all-pairs sweep visit interferers in the same ascending link order through the
    same fmaf: every output must agree bit for bit (and match the oracle)."""
    b, rbs, cues, dues = shape
    sim, pos, raw = _batch(b, rbs, cues, dues, rng_seed=sum(shape) + reward)
    h = sim.handle
    h.set_obs_mode(native.OBS_TABLE)
    h.set_reward(reward, {1: 0.0, 2: -70.0, 3: 0.0}[reward])
    snaps = {}
    for name, bucket in (('mask_walk', True), ('all_pairs', False)):
        h.set_bucketing(bucket)
        sim.step_arrays(raw)
        snaps[name] = _snapshot(sim, native, False)
    for buf, ref in snaps['mask_walk'].items():
        assert np.array_equal(snaps['all_pairs'][buf], ref), buf
    ids, cfgs, is_bs = orc.device_configs(cues, dues)
    tx, rx, ty = default_links(cues, dues)
    ref = orc.full_step(pos.astype(np.float64), tx, rx, ty, raw, orc.device_columns(cfgs, is_bs), orc.PathLossSpec(),
                        with_obs=False, chunk=8)
    assert rel_err(snaps['mask_walk']['BUF_SINR_DB'], ref['sinr_db']) <= TOL
    assert rel_err(snaps['mask_walk']['BUF_CAPACITY'], ref['capacity_mbps']) <= TOL
    if reward == 1:
        assert rel_err(snaps['mask_walk']['BUF_REWARD'][:, 0], ref['reward']) <= TOL
    elif reward == 2:
        assert rel_err(snaps['mask_walk']['BUF_REWARD'], orc.reward_shannon(ref['sinr_db'])) <= TOL
    else:
        assert rel_err(snaps['mask_walk']['BUF_REWARD'], orc.reward_cue_sinr_shannon(ref['sinr_db'], ref['rb'], ty)) <= TOL
    sim.handle.close()


@pytest.mark.parametrize('shape', [(13, 25, 25, 25), (7, 5, 9, 10), (10, 3, 30, 37), (6, 16, 50, 50)])
def test_envs_per_workgroup_and_fused_obs_are_bit_identical(native, shape):
    """Small envs share a workgroup and the LinearObs expansion may run inside the step launch: neither changes a bit
    of any output, for batch sizes that do not divide by the envs per workgroup and for odd N (8-byte obs stores)."""
    b, rbs, cues, dues = shape
    sim, pos, raw = _batch(b, rbs, cues, dues, rng_seed=sum(shape))
    h = sim.handle
    h.set_obs_mode(native.OBS_LINEAR)
    ref = None
    for bucket in (True, False):
        h.set_bucketing(bucket)
        for epw, fuse, block in ((1, 0, 0), (1, 1, 0), (2, 1, 0), (4, 0, 0), (4, 1, 512), (3, 1, 1024), (0, -1, 0)):
            if epw * ((cues + dues + 63) // 64) * 64 > 1024:
                continue
            h.set_tuning(native.TUNE_STEP_ENVS_PER_WG, epw)
            h.set_tuning(native.TUNE_STEP_FUSE_OBS, fuse)
            h.set_tuning(native.TUNE_STEP_BLOCK, block)
            h.upload(native.BUF_OBS, np.full((b, cues + dues, 6 * (cues + dues)), np.nan, np.float32))
            sim.step_arrays(raw)
            snap = _snapshot(sim, native, True)
            if ref is None:
                ref = snap
                want = orc.full_step(pos.astype(np.float64), *default_links(cues, dues), raw,
                                     orc.device_columns(*orc.device_configs(cues, dues)[1:]), orc.PathLossSpec())
                assert rel_err(snap['BUF_OBS'], want['obs']) <= TOL
                assert (snap['BUF_OBS'] == orc.expand_obs(snap['BUF_OBS_TABLE'])).all()
            for buf, r in ref.items():
                assert np.array_equal(snap[buf], r, equal_nan=True), (bucket, epw, fuse, block, buf)
    sim.handle.close()


@pytest.mark.parametrize('shape', [(9, 16, 100, 100), (5, 256, 256, 256), (7, 3, 33, 90)])
@pytest.mark.parametrize('reward', [1, 2, 3])
def test_two_links_per_thread_matches_one(native, shape, reward):
    """D2D_TUNE_STEP_LPT = 2 (links lt and lt + tpe of an env in one thread's registers, half the waves per env): every
    per-link output is bit-identical to the one-link-per-thread kernel; the SystemCapacity reward differs only by the
    order in which the capacities are summed (different wave partition), i.e. in its last bits."""
    b, rbs, cues, dues = shape
    sim, pos, raw = _batch(b, rbs, cues, dues, rng_seed=sum(shape) + reward)
    h = sim.handle
    h.set_obs_mode(native.OBS_TABLE)
    h.set_reward(reward, {1: 0.0, 2: -70.0, 3: 0.0}[reward])
    snaps = {}
    for lpt in (1, 2):
        h.set_tuning(native.TUNE_STEP_LPT, lpt)
        sim.step_arrays(raw)
        snaps[lpt] = _snapshot(sim, native, False)
    for buf, ref in snaps[1].items():
        if buf == 'BUF_REWARD' and reward == 1:
            assert np.allclose(snaps[2][buf], ref, rtol=1e-6, atol=1e-7)
        else:
            assert np.array_equal(snaps[2][buf], ref), buf
    sim.handle.close()


@pytest.mark.parametrize('shape', [(6, 64, 64, 64), (5, 128, 128, 128), (4, 96, 192, 192), (3, 300, 512, 512)],
                         ids=['128_links_one_wave', 'classes_of_128', 'classes_of_192_per_lane_records', '1024_links'])
@pytest.mark.parametrize('mode', ['table_nt', 'table_export_plain', 'none_per_env'])
def test_two_adjacent_links_per_thread_in_the_rollout_kernel(native, shape, mode):
    """csrc/d2d_rollout.hip with LPT = 2: thread t carries links 2t and 2t + 1 - 8-byte action loads and plane stores, the
    capacity tree of the one-link kernels rebuilt from the lane's own pair (the REWARD bits too are those of one link per
    thread), the table rows through LDS as 1024-byte store instructions (simulator.py:89-154, reward_fn.py:27-44,
    obs_fn.py:55-61).  Held to one link per thread and to the all-pairs sweep bit for bit, to the oracle at 1e-5, with
    nontemporal and plain stores, with and without the decoded planes, and in the obs-less per-env-reward mode."""
    b, rbs, cues, dues = shape
    sim, pos, raw = _batch(b, rbs, cues, dues, rng_seed=sum(shape))
    raw[0, :12] = raw[0, 0]                                         # twelve links on one RB: the overflow pool, two links per thread
    h = sim.handle
    h.set_obs_mode(native.OBS_NONE if mode == 'none_per_env' else native.OBS_TABLE)
    h.set_export_actions(mode == 'table_export_plain')
    h.set_tuning(native.TUNE_STEP_NT_RESULTS, 0 if mode == 'table_export_plain' else 1)
    if mode == 'none_per_env':
        h.set_reward_layout(native.REWARD_PER_ENV)
    bufs = [n for n in OUTS if not (mode == 'none_per_env' and n in ('BUF_OBS_TABLE', 'BUF_REWARD')) and not (mode != 'table_export_plain' and n in ('BUF_RB', 'BUF_PWR'))]
    if mode == 'none_per_env':
        bufs.append('BUF_REWARD_ENV')                               # the scalar once per env (d2d_set_reward_layout)
    snaps = {}
    for name, bucket, walk, lpt in (('all_pairs', False, 0, -1), ('one_link', True, 2, 1), ('two_links', True, 2, 2), ('auto', True, -1, -1)):
        h.set_bucketing(bucket)
        h.set_tuning(native.TUNE_STEP_WALK, walk)
        h.set_tuning(native.TUNE_STEP_LPT, lpt)
        if 'BUF_OBS_TABLE' in bufs:
            h.upload(native.BUF_OBS_TABLE, np.full((b, cues + dues, 6), np.nan, np.float32))
        sim.step_arrays(raw)
        snaps[name] = {n: sim.fetch(getattr(native, n)).copy() for n in bufs}
    _same(snaps, 'all_pairs')
    tx, rx, ty = default_links(cues, dues)
    ref = orc.full_step(pos.astype(np.float64), tx, rx, ty, raw, orc.device_columns(*orc.device_configs(cues, dues)[1:]), orc.PathLossSpec())
    got = snaps['two_links']
    assert rel_err(got['BUF_SINR_DB'], ref['sinr_db']) <= TOL and rel_err(got['BUF_CAPACITY'], ref['capacity_mbps']) <= TOL
    rew = got['BUF_REWARD_ENV'] if mode == 'none_per_env' else got['BUF_REWARD'][:, 0]
    assert rel_err(np.asarray(rew).reshape(b), ref['reward']) <= TOL
    if 'BUF_OBS_TABLE' in bufs:
        assert rel_err(got['BUF_OBS_TABLE'], ref['table']) <= TOL
    h.close()


def test_action_decode_is_exact_for_every_magnitude(native):
    """rb = a // P, pwr = a % P by one multiply-high below the bound stored with the magic, by division above it and for
    negatives (Python floor semantics): checked at the bound's edges, at 2^24, near 2^31 and below zero, for the CUE (24
    levels) and DUE (21 levels) alphabets."""
    import torch
    from gym_d2d_amd.envs import VecD2DEnv
    from gym_d2d_amd.envs.obs_fn import OwnLinkObsFunction
    C, P, R = 64, 64, 8                                            # N = 128: the one-env-per-workgroup kernels
    env = VecD2DEnv({'num_rbs': R, 'num_cues': C, 'num_due_pairs': P, 'obs_fn': OwnLinkObsFunction}, num_envs=3)
    env.reset(seed=3)
    edge = [0, 1, 20, 21, 23, 24, R * 21 - 1, R * 24 - 1, 65535, 65536, (1 << 24) - 2, (1 << 24) - 1, 1 << 24, (1 << 24) + 1,
            (1 << 24) + 23, 178956970, 178956971, 204522252, 204522253, (1 << 31) - 1, (1 << 31) - 24, -1, -2, -21, -24, -25,
            -(1 << 24), -(1 << 31) + 1]
    rng = np.random.default_rng(0)
    acts = rng.choice(edge, size=(3, C + P)).astype(np.int64)
    acts[0, :len(edge)] = edge; acts[1, C:C + len(edge)] = edge
    _, _, _, info = env.step(torch.as_tensor(acts.astype(np.int32), device=env.device))
    torch.cuda.synchronize()
    levels = np.array([24] * C + [21] * P)
    assert np.array_equal(info['rb'].cpu().numpy(), acts // levels[None, :])
    assert np.array_equal(info['tx_pwr_dbm'].cpu().numpy(), acts % levels[None, :])
    env.close()


@pytest.mark.parametrize('shape', [(37, 64, 64, 16, 'agent'), (64, 25, 25, 25, 'traffic')])
def test_rollout_specialisations_are_bit_identical_to_the_generic_kernel(native, shape):
    """The compile-time specialisations of the rollout configuration (HOT level 1: one env per workgroup; level 2: small
    envs sharing a workgroup with the fused LinearObs expansion) against the generic kernel with every choice made at
    run time (selected here by switching the action prefetch off, one of the conditions of the specialisations):
    every output of several steps, bit for bit."""
    import torch
    from gym_d2d_amd.envs import VecD2DEnv
    from gym_d2d_amd.envs.obs_fn import LinearObsFunction, OwnLinkObsFunction
    B, C, P, R, cue = shape
    obs_fn = OwnLinkObsFunction if cue == 'agent' else LinearObsFunction
    outs = {}
    for prefetch in (-1, 0):
        env = VecD2DEnv({'num_rbs': R, 'num_cues': C, 'num_due_pairs': P, 'obs_fn': obs_fn}, num_envs=B, cue_actions=cue)
        env.reset(seed=21)
        env.simulator.handle.set_tuning(native.TUNE_STEP_PREFETCH, prefetch)
        g = torch.Generator(device=env.device).manual_seed(3)
        snaps = []
        for k in range(3):
            act = torch.randint(0, R * 21, (B, env.num_agents), device=env.device, generator=g, dtype=torch.int32)
            obs, rew, _, info = env.step(act)
            torch.cuda.synchronize()
            snaps.append({n: v.clone() for n, v in dict(info, rew=rew, obs=obs, table=env._t['table'], flags=env._t['env_flags']).items()
                          if torch.is_tensor(v)})
        outs[prefetch] = snaps
        env.close()
    for k in range(3):
        for n, v in outs[-1][k].items():
            w = outs[0][k][n]
            same = torch.equal(v, w) if not v.is_floating_point() else torch.equal(v.view(torch.int32), w.view(torch.int32))
            assert same, (k, n)


# shapes: lists never overflow / a few envs overflow / every env overflows (25, 18 links per RB) / N > 512 (two links per
# thread) / one link per RB / a single RB
# (envs, RBs, CUEs, DUE pairs).  130, 191, 129 and 1023 links: the rollout kernel padded to the next multiple of 64 (a crowded one,
# two sparse ones at the edges of its range, the largest)
SHAPES = [(33, 25, 25, 25), (64, 256, 256, 256), (16, 4, 40, 60), (8, 7, 0, 130), (5, 300, 100, 91), (9, 1, 64, 64),
          (6, 64, 300, 400), (3, 500, 600, 600), (7, 40, 64, 65), (3, 256, 511, 512)]


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


def test_zero_distance_is_the_same_in_every_power_law_mode(native):
    """Coincident interacting devices: the reference raises ValueError('math domain error') (path_loss.py:66); the kernels
    flag the env.  What the flagged links hold (+/-inf, NaN) must not depend on WHICH power-law kernel ran: the general
    one (ple 3.5, head / tail exponent) used to turn log2(0) * tail into a NaN gain where the 1 / d^2 kernel has +inf, which
    changed the env's reward class (the ticket epilogue's inf / NaN arms)."""
    from gym_d2d_amd.path_loss import LogDistancePathLoss
    from gym_d2d_amd.simulator import Simulator
    cues, dues, rbs = 2, 2, 1
    pos = random_layout(np.random.default_rng(0), 3, cues, dues)
    pos[0, 4] = pos[0, 1]          # env 0: due00's receiver (device 4) sits on cue00 (device 1): an INTERFERER at distance 0
    pos[1, 4] = pos[1, 3]          # env 1: due00's receiver sits on its own transmitter: the SIGNAL path at distance 0
    raw = np.zeros((3, 4), np.int32) + 5          # everyone on RB 0            (env 2: nothing coincides)
    res = {}
    for ple in (2.0, 3.5):
        class Ple(LogDistancePathLoss):
            def __init__(self, f, _ple=ple):
                super().__init__(f, ple=_ple)
        sim = Simulator(dict(num_rbs=rbs, num_cues=cues, num_due_pairs=dues, num_envs=3, path_loss_model=Ple))
        sim.set_positions(pos)
        sim.set_links(sim.default_link_keys())
        sim.step_arrays(raw)
        flags = sim.fetch(native.BUF_ENV_FLAGS)
        assert (flags[:2] & native.FLAG_ZERO_DISTANCE).all() and (flags[:2] & native.FLAG_NON_FINITE).all() and flags[2] == 0
        with pytest.raises(ValueError):
            sim.check_flags()
        res[ple] = {w: sim.fetch(getattr(native, w)).copy() for w in ('BUF_SINR_DB', 'BUF_SNR_DB', 'BUF_RATE_BPS', 'BUF_CAPACITY', 'BUF_REWARD')}
        sim.handle.close()
    for w in res[2.0]:
        a, b = res[2.0][w], res[3.5][w]
        assert np.array_equal(np.isnan(a), np.isnan(b)) and np.array_equal(np.isposinf(a), np.isposinf(b)) and \
            np.array_equal(np.isneginf(a), np.isneginf(b)), (w, a, b)
        assert np.isfinite(a[2]).all() and np.isfinite(b[2]).all()
    assert not np.isfinite(res[3.5]['BUF_SINR_DB'][0, 2]) and not np.isfinite(res[3.5]['BUF_SINR_DB'][1, 2])


def test_lists_are_not_taken_automatically_where_they_cannot_help(native):
    """More than four links per RB on average at N > 1024: the automatic choice goes straight to the sweep (a ninth link on
    some RB is the rule); results equal the forced member lists (which overflow and sweep) and the oracle."""
    b, rbs, cues, dues = 2, 25, 600, 600
    sim, pos, raw = _batch(b, rbs, cues, dues, rng_seed=9)
    h = sim.handle
    h.set_obs_mode(native.OBS_TABLE)
    sim.step_arrays(raw)
    auto = {w: sim.fetch(getattr(native, w)).copy() for w in ('BUF_SINR_DB', 'BUF_CAPACITY', 'BUF_REWARD')}
    h.set_tuning(native.TUNE_STEP_WALK, 2)
    sim.step_arrays(raw)
    for w, a in auto.items():
        assert np.array_equal(sim.fetch(getattr(native, w)), a, equal_nan=True), w
    want = orc.full_step(pos.astype(np.float64), *default_links(cues, dues), raw,
                         orc.device_columns(*orc.device_configs(cues, dues)[1:]), orc.PathLossSpec(), with_obs=False)
    assert rel_err(auto['BUF_SINR_DB'], want['sinr_db']) <= TOL
    sim.handle.close()


def test_rollout_kernel_serves_the_obs_less_mode(native):
    """HOT level 1 (one env per workgroup, N a multiple of 64) with D2D_OBS_NONE, per-env reward and no decoded (rb, pwr)
    planes - the learner configuration - against the generic kernel (action prefetch off is one of the specialisation's
    conditions): every output bit for bit; the table buffer is never touched."""
    import torch
    from gym_d2d_amd.envs import VecD2DEnv
    from gym_d2d_amd.envs.obs_fn import SignalPlanesObsFunction
    B, C, P, R = 41, 64, 64, 16
    outs = {}
    for prefetch in (-1, 0):
        env = VecD2DEnv({'num_rbs': R, 'num_cues': C, 'num_due_pairs': P, 'obs_fn': SignalPlanesObsFunction}, num_envs=B,
                        export_actions=False, reward_per_env=True)
        h = env.simulator.handle
        canary = torch.full((B * (C + P) * 6,), 7.5, dtype=torch.float32, device=env.device)
        h.bind_buffer(native.BUF_OBS_TABLE, canary.data_ptr(), canary.numel() * 4)
        env.reset(seed=21)
        h.set_tuning(native.TUNE_STEP_PREFETCH, prefetch)
        g = torch.Generator(device=env.device).manual_seed(3)
        snaps = []
        for k in range(3):
            act = torch.randint(0, R * 21, (B, C + P), device=env.device, generator=g, dtype=torch.int32)
            (sinr, snr), rew, _, info = env.step(act)
            torch.cuda.synchronize()
            snaps.append({n: v.clone() for n, v in dict(info, rew=rew, sinr=sinr, snr=snr, flags=env._t['env_flags']).items() if torch.is_tensor(v)})
        assert (canary == 7.5).all()
        outs[prefetch] = snaps
        env.close()
    for k in range(3):
        for n, v in outs[-1][k].items():
            w = outs[0][k][n]
            assert torch.equal(v.view(torch.int32) if v.is_floating_point() else v, w.view(torch.int32) if w.is_floating_point() else w), (k, n)
    assert tuple(outs[-1][0]['rew'].shape) == (B,)


@pytest.mark.parametrize('shape', [(32, 64, 64), (64, 128, 128)], ids=['classes_of_64', 'classes_of_128'])
@pytest.mark.parametrize('lpt', [1, 2])
def test_rollout_kernel_rare_paths_are_bit_identical(native, lpt, shape):
    """csrc/d2d_rollout.hip keeps every per-lane rarity behind ballot branches the common wave never enters; this drives each of
    them on purpose and holds the result to the mask walk and the all-pairs sweep, bit for bit:
      env 0  three links on one RB, one interferer 0.3 m from a receiver, another 480 m away at 0 dBm: the lane's terms span more
             than the 25-bit exactness window -> its sum is re-done in sorted order;
      env 1  twelve links on one RB (four of them in the overflow pool) with the same near / far pair -> selection in ascending order;
      env 2  forty links on one RB with the pair -> more than 32 members and inexact -> the all-pairs sweep;
      env 3  two links on the same OUT-OF-RANGE RB (they interfere with each other, d2d_env.py:94-96) and a negative action;
      env 4  nine links on one RB, ordinary geometry: the pool, order-free and exact.
    Two links per thread (adjacent links 2t, 2t + 1) read the 128 records of a wave with one scalar load where the device
    classes fill aligned groups of 128 links (128 + 128), and per lane where they do not (64 + 64)."""
    rbs, cues, dues = shape
    b = 12
    sim, pos, raw = _batch(b, rbs, cues, dues, rng_seed=77)
    h = sim.handle
    n = cues + dues
    pc, pd = sim.config.num_pwr_actions['cue'], sim.config.num_pwr_actions['due']
    levels = np.array([pc] * cues + [pd] * dues)
    due_tx = lambda p: 1 + cues + 2 * p
    due_rx = lambda p: 2 + cues + 2 * p

    def crowd(env, rb, links, pwr=10):
        # everyone else off `rb`, then `links` onto it
        cur = raw[env] // levels
        raw[env] = np.where(cur == rb, ((rb + 1) % rbs) * levels + raw[env] % levels, raw[env])
        for l in links:
            raw[env, l] = rb * levels[l] + pwr

    def near_far(env, rb):
        # DUE pair 9's transmitter 0.3 m from DUE pair 5's receiver (20 dBm), CUE 3 at the cell edge with 0 dBm
        pos[env, due_tx(9)] = pos[env, due_rx(5)] + np.array([0.3, 0.0], dtype=np.float32)
        pos[env, due_rx(9)] = pos[env, due_tx(9)] + np.array([3.0, 4.0], dtype=np.float32)
        pos[env, 1 + 3] = (480.0, 0.0)
        raw[env, cues + 9] = rb * pd + 20
        raw[env, 3] = rb * pc + 0
    crowd(0, 7, [cues + 5, cues + 9, 3]); near_far(0, 7)
    crowd(1, 2, [cues + 5, cues + 9, 3] + list(range(10, 19))); near_far(1, 2)
    crowd(2, 4, [cues + 5, cues + 9, 3] + list(range(10, 47))); near_far(2, 4)
    raw[3, 11] = rbs * pc + 5; raw[3, cues + 20] = rbs * pd + 5 + (rbs * pc + 5) // pc * pd - rbs * pd   # both decode to rb = rbs (out of range)
    raw[3, 30] = -7
    crowd(4, 9, list(range(20, 29)))
    assert raw[3, 11] // pc == raw[3, cues + 20] // pd >= rbs
    sim.set_positions(pos)
    h.set_obs_mode(native.OBS_TABLE)
    snaps = {}
    for name, bucket, walk in (('mask_walk', True, 0), ('rollout', True, 2), ('all_pairs', False, 0)):
        h.set_bucketing(bucket)
        h.set_tuning(native.TUNE_STEP_WALK, walk)
        h.set_tuning(native.TUNE_STEP_LPT, lpt if name == 'rollout' else -1)
        sim.step_arrays(raw)
        snaps[name] = _snapshot(sim, native)
    _same(snaps)
    assert snaps['rollout']['BUF_ENV_FLAGS'][3] & native.FLAG_RB_OUT_OF_RANGE
    # ... and all of it against the oracle
    tx, rx, ty = default_links(cues, dues)
    cols = orc.device_columns(*orc.device_configs(cues, dues)[1:])
    ref = orc.full_step(pos.astype(np.float64), tx, rx, ty, raw, cols, orc.PathLossSpec())
    assert rel_err(snaps['rollout']['BUF_SINR_DB'], ref['sinr_db']) <= TOL
    assert rel_err(snaps['rollout']['BUF_REWARD'][:, 0], ref['reward']) <= TOL
    h.close()


@pytest.mark.parametrize('lpt', [1, 2])
def test_rollout_kernel_zero_distance_own_link_matches_the_generic_kernels(native, lpt):
    """A link whose transmitter and receiver coincide (1 / d^2: own term +inf): the rollout kernel opens its interference sum at
    minus the own term, and -inf + inf would be NaN where the kernels that leave the own link out keep the interferers' finite
    sum.  Any non-finite term sends the lane to the sorted sum, which excludes the own
    entry: same bits as the mask walk and the all-pairs sweep, alone on its RB (env 0) and with company (env 1); an INTERFERER
    on top of a receiver (env 2: accumulator inf, SINR -inf) likewise.  All three raise FLAG_ZERO_DISTANCE."""
    rbs, cues, dues = 64, 128, 128
    b = 4
    sim, pos, raw = _batch(b, rbs, cues, dues, rng_seed=123)
    h = sim.handle
    pc, pd = sim.config.num_pwr_actions['cue'], sim.config.num_pwr_actions['due']
    levels = np.array([pc] * cues + [pd] * dues)
    due_tx = lambda p: 1 + cues + 2 * p
    due_rx = lambda p: 2 + cues + 2 * p

    def alone(env, rb, links):
        cur = raw[env] // levels
        raw[env] = np.where(cur == rb, ((rb + 1) % rbs) * levels + raw[env] % levels, raw[env])
        for l in links:
            raw[env, l] = rb * levels[l] + 10
    for env in (0, 1):
        pos[env, due_rx(7)] = pos[env, due_tx(7)]
    alone(0, 5, [cues + 7])
    alone(1, 5, [cues + 7, cues + 30, 4])
    pos[2, due_tx(11)] = pos[2, due_rx(3)]                    # an interferer exactly on a receiver
    alone(2, 9, [cues + 3, cues + 11])
    sim.set_positions(pos)
    h.set_obs_mode(native.OBS_TABLE)
    snaps = {}
    for name, bucket, walk in (('mask_walk', True, 0), ('rollout', True, 2), ('all_pairs', False, 0)):
        h.set_bucketing(bucket)
        h.set_tuning(native.TUNE_STEP_WALK, walk)
        h.set_tuning(native.TUNE_STEP_LPT, lpt if name == 'rollout' else -1)
        sim.step_arrays(raw)
        snaps[name] = _snapshot(sim, native)
    _same(snaps)
    got = snaps['rollout']
    # (what the values ARE matters less than that every kernel produces the same ones: the Newton step of the SINR quotient turns
    # inf / x and x / inf into NaN everywhere; the reference raises ValueError('math domain error') at such a layout, path_loss.py:66)
    for env, link in ((0, cues + 7), (1, cues + 7), (2, cues + 3)):
        assert not np.isfinite(got['BUF_SINR_DB'][env, link]), (env, got['BUF_SINR_DB'][env, link])
    assert np.isfinite(got['BUF_SINR_DB'][1, cues + 30]) and np.isfinite(got['BUF_SINR_DB'][3]).all()
    for env in (0, 1, 2):
        assert got['BUF_ENV_FLAGS'][env] & native.FLAG_ZERO_DISTANCE
    assert got['BUF_ENV_FLAGS'][3] == 0
    h.close()


@pytest.mark.parametrize('ple', [1.2, 2.5, 3.0, 3.5, 4.4, 5.6, 7.9])
def test_power_law_with_one_integer_near_every_exponent(native, ple):
    """PL_POWK (round 6): every transmitter's exponent within 1/2 of ONE integer k -> (d^2)^(-k/2) by reciprocals and products times a
    short exp2(phi log2 d^2), |phi| <= 1/4.  Every k from 1 to 8 (odd ones take a v_rsq on top), the rollout kernel (sparse RBs) and the
    generic kernels (mask walk, all pairs): bit-identical among themselves, at the bar against the oracle."""
    from gym_d2d_amd import path_loss as pl

    class Ple(pl.LogDistancePathLoss):
        def __init__(self, f):
            super().__init__(f, ple=ple)
    b, rbs, cues, dues = 10, 48, 64, 64
    sim, pos, raw = _batch(b, rbs, cues, dues, rng_seed=int(ple * 10), path_loss_model=Ple)
    h = sim.handle
    h.set_obs_mode(native.OBS_TABLE)

    def run():
        sim.step_arrays(raw)
        return _snapshot(sim, native)
    snaps = _variants(native, h, run)
    _same(snaps)
    assert snaps['auto']['BUF_ENV_FLAGS'].max() == 0
    tx, rx, ty = default_links(cues, dues)
    cols = orc.device_columns(*orc.device_configs(cues, dues)[1:])
    ref = orc.full_step(pos.astype(np.float64), tx, rx, ty, raw, cols, orc.PathLossSpec('log_distance', 2.1, ple=ple), with_obs=False)
    for f, buf in (('sinr_db', 'BUF_SINR_DB'), ('snr_db', 'BUF_SNR_DB'), ('capacity_mbps', 'BUF_CAPACITY')):
        assert rel_err(snaps['auto'][buf], ref[f]) <= TOL, (ple, f, rel_err(snaps['auto'][buf], ref[f]))
    h.close()


def test_power_law_exponents_without_a_common_integer_keep_the_general_split(native, tmp_path):
    """COST-Hata with a base station 60 m up (slope 33.3 -> exponent 3.33, k = 3) beside UEs at 1.5 m (4.375, k = 4): uplinks and
    downlinks in one step have no common k, so the kernels keep the general split (pow_neg_half)."""
    import json
    from gym_d2d_amd import path_loss as pl
    from gym_d2d_amd.simulator import Simulator
    cfg = {'mbs': {'position': [0.0, 0.0], 'config': {'antenna_height_m': 60.0}}}
    path = tmp_path / 'tall_bs.json'
    path.write_text(json.dumps(cfg))
    b, rbs, cues, dues = 6, 8, 12, 12
    sim = Simulator(dict(num_rbs=rbs, num_cues=cues, num_due_pairs=dues, num_envs=b, path_loss_model=pl.CostHataPathLoss, device_config_file=path))
    rng = np.random.default_rng(4)
    pos = random_layout(rng, b, cues, dues)
    sim.set_positions(pos)
    keys = [('mbs', f'cue{k:02d}') for k in range(4)] + [(f'cue{k:02d}', 'mbs') for k in range(4, cues)] + list(sim.devices.dues.keys())
    sim.set_links(keys)
    n = len(keys)
    rb = rng.integers(0, rbs, (b, n)); rb[:, :4] = np.arange(4) % 2; rb[:, 4:cues] = 2 + rng.integers(0, rbs - 2, (b, cues - 4))   # up- and downlinks on disjoint RBs
    pw = rng.integers(0, 20, (b, n))
    h = sim.handle
    h.set_obs_mode(native.OBS_TABLE)

    def run():
        sim.step_arrays(rb=rb, pwr=pw)
        return _snapshot(sim, native)
    snaps = _variants(native, h, run)
    _same(snaps)
    ids, cfgs, is_bs = orc.device_configs(cues, dues)
    cfgs[0] = dict(cfgs[0], antenna_height_m=60.0)
    cols = orc.device_columns(cfgs, is_bs)
    ref = orc.step(pos.astype(np.float64), sim.link_tx, sim.link_rx, rb, pw, cols, orc.PathLossSpec('cost_hata', 2.1, area='suburban'))
    assert rel_err(snaps['auto']['BUF_SINR_DB'], ref['sinr_db']) <= TOL
    h.close()


def test_link_budget_constants_outside_the_float32_range_are_refused(native, tmp_path):
    """The kernels work in linear float32: COST-Hata's mobile-height correction (path_loss.py:105-112) applied to a RECEIVING base
    station 250 m up is a -734 dB term, 10^73 as a factor.  The library says so (D2D_ERR_UNSUPPORTED, naming the device) instead of
    returning inf / NaN planes; the same model with the station at 60 m runs (test above)."""
    import json
    from gym_d2d_amd import path_loss as pl
    from gym_d2d_amd.simulator import Simulator
    path = tmp_path / 'very_tall_bs.json'
    path.write_text(json.dumps({'mbs': {'position': [0.0, 0.0], 'config': {'antenna_height_m': 250.0}}}))
    sim = Simulator(dict(num_rbs=4, num_cues=3, num_due_pairs=3, num_envs=2, path_loss_model=pl.CostHataPathLoss, device_config_file=path))
    sim.set_positions(random_layout(np.random.default_rng(1), 2, 3, 3))
    sim.set_links(sim.default_link_keys())
    with pytest.raises(native.NativeError, match='float32 linear range') as err:
        sim.step_arrays(np.zeros((2, 6), dtype=np.int32))
    assert err.value.code == native.ERR_UNSUPPORTED
    sim.handle.close()
