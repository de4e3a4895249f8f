"""Options and entry points of the C ABI beyond the plain step: BASELINE config 2 exactly as stated, fixed (traffic-model) actions,
d2d_step_host, the export switch, the per-env reward layout, the obs-less learner configuration, the single-rank RCCL gather,
guard words around every bound buffer, gym.make, the golden cases in those modes."""
from pathlib import Path

import numpy as np
import pytest

from golden_util import load_case, rel_err
from oracle import d2d_oracle as orc
from sim_util import OUTS, assert_same as _same, default_links, random_batch as _batch, random_layout, search_variants as _variants, snapshot as _snapshot

ROOT = Path(__file__).resolve().parent.parent
pytestmark = pytest.mark.gpu
TOL = 1e-5


def test_baseline_config_2_exactly_as_stated(native):
    """BASELINE.json configs[1]: 1024 envs x (25 CUE + 25 DUE pairs, 25 RB), LogDistance, UplinkTrafficModel-driven CUEs
    (traffic_model.py:15-22): agents supply DUE actions only.  Every output of every env against the oracle."""
    import torch
    from gym_d2d_amd.envs import VecD2DEnv
    b, c, p, r = 1024, 25, 25, 25
    env = VecD2DEnv({'num_rbs': r, 'num_cues': c, 'num_due_pairs': p}, num_envs=b, cue_actions='traffic')
    obs = env.reset(seed=2024)
    assert env.num_agents == p and tuple(obs.shape) == (b, 50, 300) and tuple(env.action_buffer().shape) == (b, p)
    pos = env.simulator.positions().astype(np.float64)
    ids, cfgs, is_bs = orc.device_configs(c, p)
    cols = orc.device_columns(cfgs, is_bs)
    tx, rx, ty = default_links(c, p)
    rng = np.random.default_rng(5)
    to_np = lambda t: t.cpu().numpy() if torch.is_tensor(t) else np.asarray(t)
    for k in range(3):
        due = rng.integers(0, r * 21, (b, p)).astype(np.int32)
        act = torch.as_tensor(due, device=env.device) if env.use_torch else due
        obs, rew, dones, info = env.step(act)
        rb = np.concatenate([np.tile(np.arange(c) % r, (b, 1)), due // 21], axis=1)
        pwr = np.concatenate([np.full((b, c), 23), due % 21], axis=1)
        assert (to_np(info['rb']) == rb).all() and (to_np(info['tx_pwr_dbm']) == pwr).all()
        st = orc.step(pos, tx, rx, rb, pwr, cols, orc.PathLossSpec(), chunk=64)
        for f in ('sinr_db', 'snr_db', 'rate_bps', 'capacity_mbps'):
            assert rel_err(to_np(info[f]), st[f]) <= TOL, (k, f)
        reward = orc.reward_system_capacity(st['capacity_mbps'], rb, ty)
        assert rel_err(to_np(rew), np.repeat(reward[:, None], 50, 1)) <= TOL
        table = orc.obs_table(pos, tx, rx, st['sinr_db'], st['snr_db'])
        got_obs = to_np(obs)
        assert rel_err(got_obs, orc.expand_obs(table)) <= TOL
        assert (got_obs == orc.expand_obs(to_np(env._t['table']) if env.use_torch else env.simulator.fetch(native.BUF_OBS_TABLE))).all()
    assert env.status_flags() == 0
    env.close()


def test_fixed_actions_take_any_power_and_skip_the_decode(native):
    """a CUE whose device_config max_tx_power_dBm is above the 24-level CUE alphabet must keep its power
    when the traffic model drives it - (rb, pwr) live in the link records, there is no a // P round trip."""
    from gym_d2d_amd.envs import VecD2DEnv
    import json, tempfile, pathlib
    over = {'cue01': {'position': [120.0, -40.0],
                      'config': {'num_subcarriers': 12, 'subcarrier_spacing_kHz': 15, 'max_tx_power_dBm': 30}}}
    with tempfile.TemporaryDirectory() as tmp:
        path = pathlib.Path(tmp) / 'cfg.json'
        path.write_text(json.dumps(over))
        env = VecD2DEnv({'num_rbs': 3, 'num_cues': 4, 'num_due_pairs': 5, 'device_config_file': path}, num_envs=8,
                        cue_actions='traffic', use_torch=False)
        env.reset(seed=3)
        due = np.random.default_rng(0).integers(0, 3 * 21, (8, 5)).astype(np.int32)
        obs, rew, dones, info = env.step(due)
        assert (info['tx_pwr_dbm'][:, :4] == [23, 30, 23, 23]).all() and (info['rb'][:, :4] == [0, 1, 2, 0]).all()
        pos = env.simulator.positions().astype(np.float64)
        ids, cfgs, is_bs = orc.device_configs(4, 5, overrides={k: v['config'] for k, v in over.items()})
        tx, rx, ty = default_links(4, 5)
        st = orc.step(pos, tx, rx, info['rb'], info['tx_pwr_dbm'], orc.device_columns(cfgs, is_bs), orc.PathLossSpec())
        assert rel_err(info['sinr_db'], st['sinr_db']) <= TOL and rel_err(info['snr_db'], st['snr_db']) <= TOL
        env.close()


def test_fixed_actions_through_the_c_abi(native):
    """d2d_set_fixed_actions directly: argument checking, compact [B, A] action layout with fixed links in the MIDDLE
    of the link list, the explicit rb/pwr form keeping the full layout, and clearing."""
    sim, pos, raw = _batch(6, 5, 4, 6, rng_seed=11)
    h = sim.handle
    with pytest.raises(native.NativeError, match='out of range'):
        h.set_fixed_actions([10], [0], [0])
    with pytest.raises(native.NativeError, match='twice'):
        h.set_fixed_actions([1, 1], [0, 0], [0, 0])
    fixed_idx, fixed_rb, fixed_pw = [1, 6, 9], [4, 0, 2], [17, 3, 40]
    h.set_fixed_actions(fixed_idx, fixed_rb, fixed_pw)
    keep = [i for i in range(10) if i not in fixed_idx]
    assert h.buffer_shape(native.BUF_ACTIONS) == (6, 7)
    sim.step_arrays(raw[:, keep])
    p = sim.config.num_pwr_actions
    levels = np.array([p['cue']] * 4 + [p['due']] * 6)
    rb, pwr = raw // levels, raw % levels
    rb[:, fixed_idx] = fixed_rb; pwr[:, fixed_idx] = fixed_pw
    assert (sim.fetch(native.BUF_RB) == rb).all() and (sim.fetch(native.BUF_PWR) == pwr).all()
    ids, cfgs, is_bs = orc.device_configs(4, 6)
    tx, rx, ty = default_links(4, 6)
    st = orc.step(pos.astype(np.float64), tx, rx, rb, pwr, orc.device_columns(cfgs, is_bs), orc.PathLossSpec())
    assert rel_err(sim.fetch(native.BUF_SINR_DB), st['sinr_db']) <= TOL
    first = sim.fetch(native.BUF_SINR_DB).copy()
    # explicit form: full [B, N] arrays, entries of fixed links are ignored
    junk_rb, junk_pw = rb.copy(), pwr.copy()
    junk_rb[:, fixed_idx] = 3; junk_pw[:, fixed_idx] = 1
    sim.step_arrays(rb=junk_rb, pwr=junk_pw)
    assert np.array_equal(sim.fetch(native.BUF_SINR_DB), first)
    h.set_fixed_actions([], [], [])
    sim.step_arrays(raw)
    assert (sim.fetch(native.BUF_RB) == raw // levels).all()
    sim.handle.close()


def test_native_rccl_allgather_single_rank(native):
    """d2d_comm_unique_id / d2d_comm_init / d2d_allgather: RCCL reached through the C ABI (dlopen), one rank - the
    gathered buffer is the sent one, on the handle's stream and on a caller's side stream."""
    import torch
    from gym_d2d_amd.envs import VecD2DEnv
    env = VecD2DEnv({'num_rbs': 4, 'num_cues': 3, 'num_due_pairs': 4}, num_envs=32)
    env.reset(seed=1)
    h = env.simulator.handle
    with pytest.raises(native.NativeError, match='d2d_comm_init'):
        h.allgather(env._t['table'].data_ptr(), env._t['table'].data_ptr(), 16)
    uid = h.comm_unique_id()
    assert len(uid) == native.UNIQUE_ID_BYTES and any(uid)
    h.comm_init(1, 0, uid)
    src = env._t['table'].clone()
    dst = torch.zeros_like(src)
    h.allgather(src.data_ptr(), dst.data_ptr(), src.numel() * 4)
    torch.cuda.synchronize()
    assert torch.equal(src, dst)
    side = torch.cuda.Stream(device=env.device)
    dst.zero_()
    side.wait_stream(torch.cuda.current_stream(env.device))
    h.allgather(src.data_ptr(), dst.data_ptr(), src.numel() * 4, side.cuda_stream)
    side.synchronize()
    assert torch.equal(src, dst)
    h.comm_destroy()
    env.close()


def test_step_gatherer_native_backend_single_rank(native):
    """StepGatherer(backend='native') = the torch-free gather path; with one rank it must reproduce the local results."""
    import os
    import torch
    import torch.distributed as dist
    from gym_d2d_amd.distributed import StepGatherer
    from gym_d2d_amd.envs import VecD2DEnv
    import socket
    with socket.socket() as sock:                      # a free port: another job on the box may hold any fixed one
        sock.bind(('127.0.0.1', 0))
        port = sock.getsockname()[1]
    dist.init_process_group('gloo', init_method=f'tcp://127.0.0.1:{port}', rank=0, world_size=1)
    try:
        env = VecD2DEnv({'num_rbs': 4, 'num_cues': 3, 'num_due_pairs': 4}, num_envs=16)
        env.reset(seed=1)
        g = StepGatherer(16, 7, env.device, backend='native', handle=env.simulator.handle)
        g.gather_positions(env._t['table'])
        act = torch.randint(0, 4 * 21, (16, 7), device=env.device, dtype=torch.int32)
        env.step(act)
        g.launch(env._t['reward'], env._t['table'])
        reward, signal = g.wait()
        torch.cuda.synchronize()
        assert torch.equal(reward, env._t['reward'][:, 0]) and torch.equal(g.table(), env._t['table'])
        env.simulator.handle.comm_destroy()
        env.close()
    finally:
        dist.destroy_process_group()


def test_step_host_returns_everything_in_one_block(native):
    """d2d_step_host (the single-env drop-in's transport) against d2d_step_rb_pwr + per-buffer downloads, for a
    fused-obs size and a two-kernel size, batch of 3."""
    for cues, dues in ((6, 7), (80, 90)):
        sim, pos, raw = _batch(3, 8, cues, dues, rng_seed=cues)
        h = sim.handle
        h.set_obs_mode(native.OBS_LINEAR)
        p = sim.config.num_pwr_actions
        levels = np.array([p['cue']] * cues + [p['due']] * dues)
        rb, pwr = (raw // levels).astype(np.int32), (raw % levels).astype(np.int32)
        sim.step_arrays(rb=rb, pwr=pwr)
        want = _snapshot(sim, native, True)
        res = h.step_host(rb, pwr)
        for key, buf in (('sinr_db', 'BUF_SINR_DB'), ('snr_db', 'BUF_SNR_DB'), ('rate_bps', 'BUF_RATE_BPS'),
                         ('capacity', 'BUF_CAPACITY'), ('reward', 'BUF_REWARD'), ('obs_table', 'BUF_OBS_TABLE'),
                         ('env_flags', 'BUF_ENV_FLAGS'), ('obs', 'BUF_OBS')):
            assert np.array_equal(res[key], want[buf]), (cues, key)
        assert (res['rb'] == rb).all() and (res['pwr'] == pwr).all()
        sim.handle.close()


def test_single_env_step_is_one_packed_round_trip(native):
    """the drop-in D2DEnv.step is ONE d2d_step_host call - one packed copy each way, one
    synchronisation - and no per-buffer download.  (Structural, not a wall-clock bound: the measured latency is
    `single_env_step_ms` in every N = 1 bench line.)"""
    from gym_d2d_amd.envs import D2DEnv
    env = D2DEnv({})
    obs = env.reset()
    acts = {k: env.action_space['due' if k.startswith('due') else 'cue'].sample() for k in obs}
    h = env.simulator.handle
    calls = {'step_host': 0, 'download': 0, 'upload': 0, 'step': 0, 'step_rb_pwr': 0, 'status_flags': 0}
    for name in calls:
        def wrap(fn, name=name):
            def inner(*a, **k):
                calls[name] += 1
                return fn(*a, **k)
            return inner
        setattr(h, name, wrap(getattr(h, name)))
    for _ in range(20):
        out = env.step(acts)
    assert calls == {'step_host': 20, 'download': 0, 'upload': 0, 'step': 0, 'step_rb_pwr': 0, 'status_flags': 0}, calls
    assert len(out[0]) == 50 and next(iter(out[0].values())).dtype == np.float64      # the reference's obs dtype
    env.close()


def test_positions_written_into_a_bound_buffer_need_positions_changed(native):
    """The step kernel reads per-link position rows derived from POS_X / POS_Y; a caller that edits a BOUND position
    tensor in place says so with d2d_positions_changed."""
    import torch
    from gym_d2d_amd.envs import VecD2DEnv
    env = VecD2DEnv({'num_rbs': 4, 'num_cues': 3, 'num_due_pairs': 4}, num_envs=8)
    env.reset(seed=1)
    act = torch.randint(0, 4 * 21, (8, 7), device=env.device, dtype=torch.int32)
    _, _, _, info = env.step(act)
    before = info['snr_db'].clone()
    env._t['pos_x'][:, 1:] *= 0.5; env._t['pos_y'][:, 1:] *= 0.5            # every UE at half its distance
    env.simulator.handle.positions_changed()
    _, _, _, info = env.step(act)
    torch.cuda.synchronize()
    # inverse-square law: uplink SNRs rise by exactly 20 log10(2) dB
    assert torch.allclose(info['snr_db'][:, :3] - before[:, :3], torch.full((8, 3), 6.0206, device=env.device), atol=1e-3)
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


def test_gym_make_builds_a_working_env(tmp_path):
    """gym.make('D2DEnv-v0', env_config=...) through a stand-in `gym` package (tests/gym_stub_util.py): registration at
    import (gym_d2d/__init__.py:8-11), a gym.Env subclass, reset / step with the reference's dict conventions."""
    from pathlib import Path
    from gym_stub_util import run_gym_make
    out = run_gym_make(tmp_path, Path(__file__).resolve().parent.parent)
    assert out['entry_point'] == 'gym_d2d_amd.envs:D2DEnv' and out['make'] == 'ok' and out['is_gym_env'] is True
    assert out['agents'] == 7 and out['obs_width'] == 42 and out['obs_space'] == [42] and out['done'] == {'__all__': False}
    assert out['info_keys'] == ['capacity_mbps', 'rate_bps', 'rb', 'sinr_db', 'snr_db', 'tx_pwr_dbm']


@pytest.mark.parametrize('shape', [(9, 25, 25, 25), (5, 16, 32, 32), (4, 256, 256, 256), (3, 64, 300, 400), (2, 8, 1200, 800)])
def test_per_env_reward_layout(native, shape):
    """D2D_REWARD_PER_ENV: SystemCapacity's scalar (reward_fn.py:42-44) once per env in D2D_BUF_REWARD_ENV, the same bits as
    column 0 of the [B, N] rows, which are then left alone - in every kernel family (small envs sharing a workgroup, one
    env per workgroup with the ticket epilogue, two links per thread, strided)."""
    b, rbs, cues, dues = shape
    n = cues + dues
    sim, pos, raw = _batch(b, rbs, cues, dues, rng_seed=sum(shape) + 1)
    h = sim.handle
    h.set_obs_mode(native.OBS_TABLE)
    sim.step_arrays(raw)
    rows = sim.fetch(native.BUF_REWARD).copy()
    assert (rows == rows[:, :1]).all()
    h.set_reward_layout(native.REWARD_PER_ENV)
    h.upload(native.BUF_REWARD, np.full((b, n), -7.0, np.float32))
    h.upload(native.BUF_REWARD_ENV, np.full((b,), np.nan, np.float32))
    sim.step_arrays(raw)
    assert np.array_equal(sim.fetch(native.BUF_REWARD_ENV), rows[:, 0])
    assert (sim.fetch(native.BUF_REWARD) == -7.0).all()                  # untouched
    # the -1 rule through the per-env route (reward_fn.py:29-41): min_capacity above every link's capacity
    h.set_reward(native.REWARD_SYSTEM_CAPACITY, 1.0e9)
    sim.step_arrays(raw)
    ty = default_links(cues, dues)[2]
    cap = sim.fetch(native.BUF_CAPACITY).astype(np.float64)
    want = orc.reward_system_capacity(cap, sim.fetch(native.BUF_RB), ty, 1.0e9)
    got = sim.fetch(native.BUF_REWARD_ENV)
    assert np.array_equal(got == -1.0, want == -1.0) and rel_err(got, want) <= 1e-6
    # the per-agent rewards ignore the layout
    h.set_reward(native.REWARD_SHANNON, -70.0)
    sim.step_arrays(raw)
    per_agent = sim.fetch(native.BUF_REWARD).copy()
    h.set_reward_layout(native.REWARD_PER_AGENT)
    sim.step_arrays(raw)
    assert np.array_equal(sim.fetch(native.BUF_REWARD), per_agent) and not (per_agent == -7.0).all()
    with pytest.raises(native.NativeError):
        h.set_reward_layout(5)
    sim.handle.close()


def test_obs_less_learner_configuration(native):
    """SignalPlanesObsFunction (D2D_OBS_NONE) + reward_per_env + export off: what a learner that builds its own features
    runs.  The planes and link_positions() reproduce the table env's observation bit for bit; StepGatherer(mode='planes')
    (one rank, gloo) assembles the same [B, N, 6] table as the table plan."""
    import socket
    import torch
    import torch.distributed as dist
    from gym_d2d_amd.distributed import StepGatherer
    from gym_d2d_amd.envs import VecD2DEnv
    from gym_d2d_amd.envs.obs_fn import OwnLinkObsFunction, SignalPlanesObsFunction
    cfg = {'num_rbs': 16, 'num_cues': 32, 'num_due_pairs': 32}
    ref = VecD2DEnv(dict(cfg, obs_fn=OwnLinkObsFunction), num_envs=40)
    env = VecD2DEnv(dict(cfg, obs_fn=SignalPlanesObsFunction), num_envs=40, export_actions=False, reward_per_env=True)
    t0 = ref.reset(seed=5)
    sinr0, snr0 = env.reset(seed=5)
    assert env._t['table'] is None
    lp = env.link_positions()
    assert tuple(lp.shape) == (40, 64, 4) and torch.equal(lp, t0[:, :, :4])
    assert torch.equal(sinr0, t0[:, :, 4]) and torch.equal(snr0, t0[:, :, 5])
    with socket.socket() as sock:
        sock.bind(('127.0.0.1', 0))
        port = sock.getsockname()[1]
    dist.init_process_group('gloo', init_method=f'tcp://127.0.0.1:{port}', rank=0, world_size=1)
    try:
        gt = StepGatherer(40, 64, ref.device)
        gp = StepGatherer(40, 64, env.device, mode='planes')
        gt.gather_positions(t0)
        gp.gather_positions(lp)
        for k in range(3):
            act = torch.randint(0, 16 * 21, (40, 64), device=env.device, dtype=torch.int32)
            table, r_ref, _, _ = ref.step(act)
            (sinr, snr), r_env, _, info = env.step(act)
            assert tuple(r_env.shape) == (40,) and torch.equal(r_env, r_ref[:, 0])
            assert torch.equal(sinr, table[:, :, 4]) and torch.equal(snr, table[:, :, 5])
            assert info['rb'] is None and info['tx_pwr_dbm'] is None
            gt.launch(r_ref, table)
            gp.launch(r_env, sinr=sinr, snr=snr)
            rt, _ = gt.wait()
            rp, planes = gp.wait()
            torch.cuda.synchronize()
            assert torch.equal(rt, rp) and torch.equal(gt.table(), gp.table()) and torch.equal(gp.table(), table)
        env.reset()                                                        # a new episode: the library refreshes the rows in place
        assert torch.equal(env.link_positions(), ref.reset()[:, :, :4])
    finally:
        dist.destroy_process_group()
    with pytest.raises(ValueError):
        VecD2DEnv(dict(cfg, reward_fn=__import__('gym_d2d_amd.envs.reward_fn', fromlist=['x']).ShannonRewardFunction), num_envs=4,
                  reward_per_env=True)
    ref.close(); env.close()


GUARD = 0x5AFEC0DE


@pytest.mark.parametrize('case', [
    dict(b=3, rbs=1, cues=1, dues=0),                                      # N = 1
    dict(b=5, rbs=7, cues=24, dues=25),                                    # N = 49: odd, 8-byte fused expansion
    dict(b=9, rbs=25, cues=25, dues=25, epw=4),                            # N = 50, envs sharing a workgroup, fused
    dict(b=9, rbs=25, cues=25, dues=25, fuse=0),                           # N = 50, stand-alone expansion
    dict(b=3, rbs=64, cues=255, dues=256),                                 # N = 511
    dict(b=3, rbs=256, cues=256, dues=256),                                # N = 512: the rollout kernel
    dict(b=2, rbs=2, cues=256, dues=256, walk=2),                          # every list overflows (256 links per RB)
    dict(b=2, rbs=700, cues=1024, dues=1024, obs='table'),                 # N = 2048
    dict(b=2, rbs=40, cues=700, dues=701, obs='table'),                    # N = 1401: odd, two links per thread
])
def test_guard_words_around_every_bound_buffer_survive(native, case):
    """Every D2D_BUF_* the step writes is bound INSIDE a larger allocation with guard words on both sides (SURVEY.md
    section 5's out-of-bounds canaries): after reset + steps in each kernel family the guards are intact and the results
    equal those of an unguarded run."""
    import torch
    from gym_d2d_amd.simulator import Simulator
    b, rbs, cues, dues = case['b'], case['rbs'], case['cues'], case['dues']
    n, d = cues + dues, 1 + cues + 2 * dues
    obs_mode = {'linear': native.OBS_LINEAR, 'table': native.OBS_TABLE}[case.get('obs', 'linear')]
    dev = torch.device('cuda', 0)

    def run(guarded):
        sim = Simulator(dict(num_rbs=rbs, num_cues=cues, num_due_pairs=dues, num_envs=b), max_links=n)
        h = sim.handle
        sim.set_links(sim.default_link_keys())
        h.set_obs_mode(obs_mode)
        for key, name in ((native.TUNE_STEP_ENVS_PER_WG, 'epw'), (native.TUNE_STEP_FUSE_OBS, 'fuse'), (native.TUNE_STEP_WALK, 'walk')):
            if name in case:
                h.set_tuning(key, case[name])
        sizes = {native.BUF_POS_X: b * d, native.BUF_POS_Y: b * d, native.BUF_ACTIONS: b * n, native.BUF_RB: b * n,
                 native.BUF_PWR: b * n, native.BUF_SINR_DB: b * n, native.BUF_SNR_DB: b * n, native.BUF_RATE_BPS: b * n,
                 native.BUF_CAPACITY: b * n, native.BUF_REWARD: b * n, native.BUF_OBS_TABLE: b * n * 6,
                 native.BUF_ENV_FLAGS: b, native.BUF_REWARD_ENV: b}
        if obs_mode == native.OBS_LINEAR:
            sizes[native.BUF_OBS] = b * n * 6 * n
        arenas = {}
        if guarded:
            for which, words in sizes.items():
                pad = 64                                                   # 256 bytes of guard on each side
                arena = torch.full((words + 2 * pad,), GUARD, dtype=torch.int32, device=dev)
                arenas[which] = (arena, pad, words)
                h.bind_buffer(which, arena.data_ptr() + pad * 4, words * 4)
        rng = np.random.default_rng(17)
        p = sim.config.num_pwr_actions
        out = []
        for layout in (native.REWARD_PER_AGENT, native.REWARD_PER_ENV):
            h.set_reward_layout(layout)
            sim.reset_device(seed=3, episode=0)
            for k in range(2):
                raw = np.concatenate([rng.integers(0, rbs * p['cue'], (b, cues)), rng.integers(0, rbs * p['due'], (b, dues))],
                                     axis=1).astype(np.int32)
                sim.step_arrays(raw)
            skip = (native.BUF_ACTIONS, native.BUF_POS_X, native.BUF_POS_Y,
                    native.BUF_REWARD_ENV if layout == native.REWARD_PER_AGENT else native.BUF_REWARD)     # not written under this layout
            out.append({w: sim.fetch(w).copy() for w in sizes if w not in skip})
        torch.cuda.synchronize()
        for which, (arena, pad, words) in arenas.items():
            host = arena.cpu().numpy().view(np.uint32)
            assert (host[:pad] == GUARD).all(), ('front guard', which)
            assert (host[pad + words:] == GUARD).all(), ('back guard', which)
        sim.handle.close()
        return out

    plain, guarded = run(False), run(True)
    for a, g in zip(plain, guarded):
        for which in a:
            assert np.array_equal(a[which], g[which], equal_nan=True), which
    assert native.BUF_REWARD in plain[0] and native.BUF_REWARD_ENV in plain[1]


def _golden_names():
    from golden_util import case_names
    return [n for n in case_names() if 'shadowing' not in n]


@pytest.mark.parametrize('name', _golden_names())
def test_golden_cases_in_the_round4_modes(native, name):
    """Every captured reference case (tests/golden, made by running the reference) through the modes round 4 added: the obs-less step
    with the per-env reward (where the library takes the member lists by itself) - SINR / SNR / rate / capacity and
    SystemCapacity's scalar within 1e-5 of the reference, the link-position rows bit-exact - and the float64 obs block of
    d2d_set_obs_dtype within 1e-5 of the reference's float64 observations."""
    from golden_util import load_case
    from sim_util import env_config_for
    from gym_d2d_amd.simulator import Simulator
    case = load_case(name)
    sim = Simulator(env_config_for(case))
    sim.set_positions(case.pos[None] if 'unrounded' in name else case.pos[None].astype(np.float32))    # case16: the reference's float64 layouts
    h = sim.handle
    for k, s in enumerate(case.steps):
        sim.set_links([tuple(key.split(':')) for key in s.keys])
        tag = (name, k)
        h.set_reward(native.REWARD_SYSTEM_CAPACITY, 0.0)
        h.set_obs_mode(native.OBS_NONE)
        h.set_reward_layout(native.REWARD_PER_ENV)
        h.set_export_actions(False)
        sim.step_arrays(rb=s.rb[None], pwr=s.pwr[None])
        assert sim.check_flags() & native.FLAG_ZERO_DISTANCE == 0
        for f, buf in (('sinr_db', native.BUF_SINR_DB), ('snr_db', native.BUF_SNR_DB), ('rate_bps', native.BUF_RATE_BPS),
                       ('capacity_mbps', native.BUF_CAPACITY)):
            assert rel_err(sim.fetch(buf)[0], getattr(s, f)) <= TOL, (tag, f)
        assert rel_err(sim.fetch(native.BUF_REWARD_ENV), np.asarray(s.reward_system_capacity).reshape(-1)[:1]) <= TOL, tag
        assert (h.download(native.BUF_LINK_POS)[0] == s.obs_table[:, :4].astype(np.float32)).all(), tag
        h.set_obs_mode(native.OBS_LINEAR)
        h.set_reward_layout(native.REWARD_PER_AGENT)
        h.set_obs_dtype(native.F64)
        sim.step_arrays(rb=s.rb[None], pwr=s.pwr[None])
        obs = sim.fetch(native.BUF_OBS)[0]
        assert obs.dtype == np.float64 and rel_err(obs[s.obs_rows], s.obs) <= TOL, tag
        h.set_obs_dtype(native.F32)
        h.set_export_actions(True)
    h.close()


def test_numpy_path_of_the_round4_options(native):
    """VecD2DEnv(use_torch=False) - host arrays in and out, the library owning every buffer - with the per-env reward, the obs-less
    observation and export off: the same numbers as the torch path."""
    import torch
    from gym_d2d_amd.envs import VecD2DEnv
    from gym_d2d_amd.envs.obs_fn import SignalPlanesObsFunction
    cfg = {'num_rbs': 8, 'num_cues': 12, 'num_due_pairs': 20, 'obs_fn': SignalPlanesObsFunction}
    t = VecD2DEnv(dict(cfg), num_envs=24, export_actions=False, reward_per_env=True)
    n = VecD2DEnv(dict(cfg), num_envs=24, export_actions=False, reward_per_env=True, use_torch=False)
    (ts, tn), (ns, nn) = t.reset(seed=8), n.reset(seed=8)
    assert isinstance(ns, np.ndarray) and np.array_equal(ts.cpu().numpy(), ns) and np.array_equal(tn.cpu().numpy(), nn)
    assert np.array_equal(t.link_positions().cpu().numpy(), n.link_positions())
    rng = np.random.default_rng(0)
    for k in range(3):
        act = rng.integers(0, 8 * 21, (24, 32)).astype(np.int32)
        (ts, tn), tr, td, ti = t.step(torch.as_tensor(act, device=t.device))
        (ns, nn), nr, nd, ni = n.step(act)
        assert nr.shape == (24,) and np.array_equal(tr.cpu().numpy(), nr) and np.array_equal(ts.cpu().numpy(), ns)
        assert ni['rb'] is None and ni['tx_pwr_dbm'] is None and np.array_equal(ti['capacity_mbps'].cpu().numpy(), ni['capacity_mbps'])
    t.close(); n.close()



def test_link_pair_table_is_the_device_table_without_the_unused_entries(native):
    """d2d_set_path_loss_link_table ([B,N,N] by (tx link, rx link): exactly the pairs the step reads) against
    d2d_set_path_loss_table ([B,D,D] by device, everything else NaN): every output bit for bit, per-env and shared, a link list
    in scrambled order; a new link list drops the table (D2D_ERR_STATE until a path-loss model is set again)."""
    from gym_d2d_amd.simulator import Simulator
    b, rbs, cues, dues = 5, 6, 7, 9
    rng = np.random.default_rng(3)
    sim = Simulator(dict(num_rbs=rbs, num_cues=cues, num_due_pairs=dues, num_envs=b))
    h = sim.handle
    pos = random_layout(rng, b, cues, dues)
    sim.set_positions(pos)
    keys = sim.default_link_keys()
    order = rng.permutation(len(keys))
    sim.set_links([keys[k] for k in order])
    n, d = len(keys), 1 + cues + 2 * dues
    raw = np.concatenate([rng.integers(0, rbs * 24, (b, cues)), rng.integers(0, rbs * 21, (b, dues))], 1).astype(np.int32)[:, order]
    p64 = pos.astype(np.float64)
    dist = np.hypot(p64[:, :, None, 0] - p64[:, None, :, 0], p64[:, :, None, 1] - p64[:, None, :, 1])
    with np.errstate(divide='ignore'):
        full = 35.0 + 27.0 * np.log10(dist)                       # [B, D, D]; the diagonal is -inf and never read
    h.set_obs_mode(native.OBS_TABLE)
    for per_env in (True, False):
        dev = full if per_env else full[0]
        used = np.full_like(dev, np.nan)
        used[..., sim.link_tx[:, None], sim.link_rx[None, :]] = dev[..., sim.link_tx[:, None], sim.link_rx[None, :]]
        h.set_path_loss_table(used)
        sim.step_arrays(raw)
        a = _snapshot(sim, native)
        h.set_path_loss_link_table(np.ascontiguousarray(dev[..., sim.link_tx[:, None], sim.link_rx[None, :]]))
        sim.step_arrays(raw)
        c = _snapshot(sim, native)
        for buf in a:
            assert np.array_equal(a[buf], c[buf], equal_nan=True), (per_env, buf)
        assert np.isfinite(c['BUF_SINR_DB']).all()
    with pytest.raises(native.NativeError):
        h.set_path_loss_link_table(np.zeros((n + 1, n + 1)))       # not the length of the link list
    h.set_links(sim.link_tx[:4], sim.link_rx[:4], sim.link_type[:4])   # a new list: the link-indexed table is gone with the old one
    with pytest.raises(native.NativeError) as exc:
        h.step()
    assert exc.value.code == native.ERR_STATE
    h.close()


def test_profile_median_beside_the_mean(native):
    """d2d_profile_median: the median launch duration of the events d2d_profile_enable brackets every launch with - what bench.py
    quotes beside the mean, which a cold first launch pulls up."""
    sim, pos, raw = _batch(64, 16, 32, 32, rng_seed=5)
    h = sim.handle
    h.set_obs_mode(native.OBS_LINEAR)
    h.set_tuning(native.TUNE_STEP_FUSE_OBS, 0)                      # two kernels per step: both are timed
    h.profile_reset(); h.profile_enable(True)
    for _ in range(25):
        sim.step_arrays(raw)
    for kernel in (0, 1):
        total, launches = h.profile_read(kernel)
        med = h.profile_median(kernel)
        assert launches == 25 and 0.0 < med <= 1.5 * total / launches, (kernel, total, launches, med)
    h.profile_enable(False)
    h.profile_reset()
    assert h.profile_median(0) == 0.0
    h.close()
