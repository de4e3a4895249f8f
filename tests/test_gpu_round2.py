"""GPU tests of the round-2 additions, through the C ABI: interferer-search variants and launch geometries are
bit-identical, BASELINE config 2 exactly as stated, traffic-model links held in the kernel's records, the gather-side
entry points (d2d_expand_table, d2d_comm_* / d2d_allgather) and the packed host step (d2d_step_host)."""
import ctypes

import numpy as np
import pytest

from golden_util import load_case, rel_err
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


def _batch(native, num_envs, rbs, cues, dues, seed, **cfg):
    from gym_d2d_amd.simulator import Simulator
    rng = np.random.default_rng(seed)
    sim = Simulator(dict(num_rbs=rbs, num_cues=cues, num_due_pairs=dues, num_envs=num_envs, **cfg))
    pos = random_layout(rng, num_envs, cues, dues)
    sim.set_positions(pos)
    sim.set_links(sim.default_link_keys())
    p = sim.config.num_pwr_actions
    raw = np.concatenate([rng.integers(0, rbs * p['cue'], (num_envs, cues)),
                          rng.integers(0, rbs * p['due'], (num_envs, dues))], axis=1).astype(np.int32)
    return sim, pos, raw


def _snapshot(sim, native, with_obs):
    out = {name: sim.fetch(getattr(native, name)).copy() for name in OUTS}
    if with_obs:
        out['BUF_OBS'] = sim.fetch(native.BUF_OBS).copy()
    return out


@pytest.mark.parametrize('shape', [(33, 25, 25, 25), (24, 256, 256, 256), (16, 4, 40, 60), (8, 7, 0, 130), (5, 300, 100, 91),
                                   (9, 1, 64, 64)])
@pytest.mark.parametrize('reward', [1, 2, 3])
def test_interferer_search_variants_are_bit_identical(native, shape, reward):
    """The bitmask walk and the masked all-pairs sweep visit interferers in the same ascending link order through the
    same fmaf: every output must agree bit for bit (and match the oracle)."""
    b, rbs, cues, dues = shape
    sim, pos, raw = _batch(native, b, rbs, cues, dues, seed=sum(shape) + reward)
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
    sim, pos, raw = _batch(native, b, rbs, cues, dues, seed=sum(shape))
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
    sim, pos, raw = _batch(native, 6, 5, 4, 6, seed=11)
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


def test_expand_table_of_a_gathered_table_is_bit_identical_to_the_local_obs(native):
    """d2d_expand_table (the learner-side expansion of tables received from other GPUs) runs the same kernel: the
    expansion of a one-rank 'gathered' table equals the D2D_BUF_OBS the owning handle produced, bit for bit - at a
    size the step fuses (N = 24) and at one it does not (N = 192), and on a handle with a different B / N."""
    import torch
    from gym_d2d_amd.distributed import expand_table, expand_table_torch
    from gym_d2d_amd.envs import VecD2DEnv
    other = VecD2DEnv({'num_rbs': 2, 'num_cues': 1, 'num_due_pairs': 1}, num_envs=2)
    for cues, dues, b in ((10, 14, 40), (96, 96, 16), (5, 6, 3)):
        env = VecD2DEnv({'num_rbs': 9, 'num_cues': cues, 'num_due_pairs': dues}, num_envs=b)
        obs = env.reset(seed=1)
        gathered = env._t['table'].clone()                  # what an all-gather with one rank delivers
        for h in (env.simulator.handle, other.simulator.handle):
            out = expand_table(gathered, h)
            torch.cuda.synchronize()
            assert torch.equal(out, obs)
        assert torch.equal(expand_table_torch(gathered.cpu()), obs.cpu())
        with pytest.raises(ValueError, match='native handle'):
            expand_table(gathered)
        env.close()
    other.close()


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
        sim, pos, raw = _batch(native, 3, 8, cues, dues, seed=cues)
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


def test_device_reset_never_places_two_interacting_devices_together(native):
    """the radius uniform is on the open interval, so no CUE lands on the BS and no DUE receiver
    on its transmitter; VecD2DEnv.reset() checks the zero-distance flag once per reset."""
    from gym_d2d_amd.envs import VecD2DEnv
    env = VecD2DEnv({'num_rbs': 8, 'num_cues': 64, 'num_due_pairs': 64}, num_envs=2048, use_torch=False)
    for ep in range(3):
        env.reset(seed=ep)
        pos = env.simulator.positions().astype(np.float64)
        r_cue = np.hypot(pos[:, 1:65, 0], pos[:, 1:65, 1])
        tx, rx = pos[:, 65::2], pos[:, 66::2]
        d_pair = np.hypot(tx[..., 0] - rx[..., 0], tx[..., 1] - rx[..., 1])
        assert r_cue.min() > 0.0 and d_pair.min() > 0.0
        assert env.status_flags() & (native.FLAG_ZERO_DISTANCE | native.FLAG_NON_FINITE) == 0
    # and the check itself: positions forced onto the base station are reported at reset-time granularity
    xy = env.simulator.positions()
    xy[5, 3] = 0.0
    env.simulator.set_positions(xy)
    env.step(np.zeros((2048, 128), np.int32))
    assert env.status_flags() & native.FLAG_ZERO_DISTANCE
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


def test_device_reset_against_the_reference_samplers(native):
    """csrc/d2d_reset.hip against positions the REFERENCE's samplers produced from the same Philox uniforms (golden
    sampler_case15): fp32 sincos / sqrt vs the reference's fp64, so 1e-6 of the cell radius; a rejection decision may
    differ only for a candidate within rounding of the cell edge."""
    import json
    from golden_util import GOLDEN_DIR
    from gym_d2d_amd.simulator import Simulator
    z = np.load(GOLDEN_DIR / 'sampler_case15.npz')
    for c in json.loads(bytes(z['meta_json']).decode())['configs']:
        ref = z[c['tag'] + '_pos']
        sim = Simulator(dict(num_cues=c['num_cues'], num_due_pairs=c['num_due_pairs'], num_envs=c['num_envs'],
                             cell_radius_m=c['cell_radius_m'], d2d_radius_m=c['d2d_radius_m']))
        sim.handle.set_env_offset(c['first_env'])
        sim.reset_device(seed=c['seed'], episode=c['episode'])
        got = sim.positions().astype(np.float64)
        close = np.abs(got - ref).max(axis=2) <= c['cell_radius_m'] * 2e-6
        assert close.mean() > 0.995, (c['tag'], close.mean())
        # the few misses must be rejection decisions at the cell edge: the kernel's own position is still legal
        assert (np.hypot(got[..., 0], got[..., 1]) <= c['cell_radius_m'] * (1 + 1e-6)).all()
        sim.handle.close()


@pytest.mark.parametrize('shape', [(9, 16, 100, 100), (5, 256, 256, 256), (7, 3, 33, 90)])
@pytest.mark.parametrize('reward', [1, 2, 3])
def test_two_links_per_thread_matches_one(native, shape, reward):
    """D2D_TUNE_STEP_LPT = 2 (links lt and lt + tpe of an env in one thread's registers, half the waves per env): every
    per-link output is bit-identical to the one-link-per-thread kernel; the SystemCapacity reward differs only by the
    order in which the capacities are summed (different wave partition), i.e. in its last bits."""
    b, rbs, cues, dues = shape
    sim, pos, raw = _batch(native, b, rbs, cues, dues, seed=sum(shape) + reward)
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


def test_reset_writes_the_link_position_rows_itself_for_the_standard_link_list(native):
    """With every uplink and sidelink in device order (the env's own list) d2d_reset_positions fills the per-link rows in
    the sampler kernel; any other list, or a caller saying positions_changed, goes through the gather kernel.  Same step
    results, bit for bit - and a reordered link list (gather route) agrees with the standard one link by link."""
    import torch
    from gym_d2d_amd.envs import VecD2DEnv
    cfg = {'num_rbs': 5, 'num_cues': 6, 'num_due_pairs': 7}
    outs = []
    for route in ('sampler', 'gather'):
        env = VecD2DEnv(dict(cfg), num_envs=33)
        env.reset(seed=77)
        if route == 'gather':
            env.simulator.handle.positions_changed()
        act = torch.randint(0, 5 * 21, (33, 13), device=env.device, dtype=torch.int32,
                            generator=torch.Generator(device=env.device).manual_seed(5))
        obs, rew, _, info = env.step(act)
        torch.cuda.synchronize()
        outs.append({k: v.clone() for k, v in dict(info, obs=obs, rew=rew).items() if torch.is_tensor(v)})
        env.close()
    for k in outs[0]:
        assert torch.equal(outs[0][k], outs[1][k]), k
    # table columns 0..3 are the link's own tx / rx coordinates: they must be the sampled device positions
    env = VecD2DEnv(dict(cfg, obs_fn=__import__('gym_d2d_amd.envs.obs_fn', fromlist=['x']).OwnLinkObsFunction), num_envs=33)
    env.reset(seed=77)
    act = torch.zeros((33, 13), device=env.device, dtype=torch.int32)
    env.step(act)
    torch.cuda.synchronize()
    t, px, py = env._t['table'].cpu().numpy(), env._t['pos_x'].cpu().numpy(), env._t['pos_y'].cpu().numpy()
    tx = np.array([i + 1 for i in range(6)] + [7 + 2 * k for k in range(7)])
    rx = np.array([0] * 6 + [8 + 2 * k for k in range(7)])
    assert np.array_equal(t[:, :, 0], px[:, tx]) and np.array_equal(t[:, :, 1], py[:, tx])
    assert np.array_equal(t[:, :, 2], px[:, rx]) and np.array_equal(t[:, :, 3], py[:, rx])
    env.close()


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
