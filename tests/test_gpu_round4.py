"""GPU tests of the round-4 additions, through the C ABI: the packed fused LinearObs expansion at every geometry, the
per-env reward layout (D2D_REWARD_PER_ENV), the link-position rows (D2D_BUF_LINK_POS), the obs-less learner configuration
(SignalPlanesObsFunction + StepGatherer(mode='planes')), zero distance under the power-law path loss, guard words around
every bound buffer, the staged write probe."""
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent

from golden_util import rel_err
from oracle import d2d_oracle as orc
from sim_util import default_links, random_layout

pytestmark = pytest.mark.gpu
TOL = 1e-5


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


# (envs, rbs, cues, dues): 6N % 4 == 0 everywhere (16-byte expansion); envs smaller than a pass (N = 2, 4), a row longer
# than a pass (N = 100, 128 at 64 / 128-thread blocks), batch sizes that leave the last workgroup partly empty
FUSED_SHAPES = [(7, 2, 1, 1), (9, 3, 2, 2), (5, 4, 3, 3), (33, 5, 4, 6), (11, 7, 10, 12), (13, 25, 25, 25), (6, 16, 32, 32),
                (5, 9, 50, 50), (3, 30, 64, 64)]


@pytest.mark.parametrize('shape', FUSED_SHAPES)
def test_packed_fused_expansion_is_bit_identical_at_every_geometry(native, shape):
    """The fused expansion walks the workgroup's contiguous obs region in passes of blockDim float4 with an incrementally
    advanced (env, row, column) per lane and a rotated start: every (envs per workgroup, block, rotate) must give the
    bits of the stand-alone expansion kernel (obs_fn.py:43-53)."""
    b, rbs, cues, dues = shape
    n = cues + dues
    sim, pos, raw = _batch(b, rbs, cues, dues, rng_seed=sum(shape))
    h = sim.handle
    h.set_obs_mode(native.OBS_LINEAR)
    h.set_tuning(native.TUNE_STEP_FUSE_OBS, 0)
    sim.step_arrays(raw)
    ref_obs = sim.fetch(native.BUF_OBS).copy()
    ref_tab = sim.fetch(native.BUF_OBS_TABLE).copy()
    assert (ref_obs == orc.expand_obs(ref_tab)).all()
    tpe = ((n + 63) // 64) * 64
    tried = 0
    for epw in (1, 2, 3, 4, 8, 16):
        if epw * tpe > 1024:
            continue
        for block in (0, 64, 128, 256, 320, 512, 1024):
            if block and block < epw * tpe:
                continue
            for rotate in (0, 29, 1):
                h.set_tuning(native.TUNE_STEP_ENVS_PER_WG, epw)
                h.set_tuning(native.TUNE_STEP_FUSE_OBS, 1)
                h.set_tuning(native.TUNE_STEP_BLOCK, block)
                h.set_tuning(native.TUNE_STEP_OBS_ROTATE, rotate)
                h.upload(native.BUF_OBS, np.full((b, n, 6 * n), np.nan, np.float32))
                sim.step_arrays(raw)
                got = sim.fetch(native.BUF_OBS)
                assert np.array_equal(got, ref_obs), (epw, block, rotate, np.argwhere(got != ref_obs)[:3])
                tried += 1
    assert tried >= 12
    sim.handle.close()


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


def test_link_position_rows_are_the_table_columns(native):
    """D2D_BUF_LINK_POS [B, N, 4] = columns 0-3 of the obs table (obs_fn.py:57-59), kept current by the library across
    set_positions, device-side resets and link-list changes; read-only."""
    b, rbs, cues, dues = 6, 5, 7, 9
    sim, pos, raw = _batch(b, rbs, cues, dues, rng_seed=3)
    h = sim.handle
    h.set_obs_mode(native.OBS_TABLE)
    rows = h.download(native.BUF_LINK_POS)                                 # before any step: derived on demand
    assert rows.shape == (b, cues + dues, 4)
    sim.step_arrays(raw)
    assert np.array_equal(rows, sim.fetch(native.BUF_OBS_TABLE)[:, :, :4])
    tx, rx, _ = default_links(cues, dues)
    assert np.array_equal(rows[:, :, :2], pos[:, tx]) and np.array_equal(rows[:, :, 2:], pos[:, rx])
    sim.reset_device(seed=11, episode=0)                                   # device-side sampler writes the rows itself
    rows2 = h.download(native.BUF_LINK_POS)
    p2 = sim.positions()
    assert np.array_equal(rows2[:, :, :2], p2[:, tx]) and np.array_equal(rows2[:, :, 2:], p2[:, rx]) and not np.array_equal(rows, rows2)
    keys = sim.default_link_keys()[::2]                                    # a sub-list of the links
    sim.set_links(keys)
    rows3 = h.download(native.BUF_LINK_POS)
    assert rows3.shape == (b, len(keys), 4) and np.array_equal(rows3, rows2[:, ::2])
    ptr, nbytes = h.get_buffer(native.BUF_LINK_POS)
    assert nbytes == b * len(keys) * 16 and ptr
    with pytest.raises(native.NativeError):
        h.bind_buffer(native.BUF_LINK_POS, ptr, nbytes)
    with pytest.raises(native.NativeError):
        h.upload(native.BUF_LINK_POS, rows3)
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


def test_staged_write_probe(native):
    """libd2d_probe.so's staged forms: the fill family with an LDS stage + barrier and / or a per-wave sleep stagger in front of
    the stores - runs, writes what it says, refuses nonsense."""
    import sys
    sys.path.insert(0, str(ROOT / 'tools'))
    import write_probe
    for variant, stagger in ((0, 0), (32, 0), (64, 2), (96, 1), (32 + 1, 0), (128, 0), (256 + 32, 0), (384, 0), (512 + 1, 0)):
        assert write_probe.write_staged(64 << 20, variant, stagger, iters=2) > 100.0
    import torch
    dst = torch.zeros(64 << 18, dtype=torch.float32, device='cuda')       # 64 MiB
    assert write_probe.write_staged(64 << 20, 32, 0, iters=1, dst_ptr=dst.data_ptr()) > 100.0
    with pytest.raises(ValueError):
        write_probe.write_staged(1 << 20, 0)
    with pytest.raises(ValueError):
        write_probe.write_staged(64 << 20, 640)


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


@pytest.mark.parametrize('shape', [(5, 7, 24, 25), (9, 25, 25, 25), (3, 256, 256, 256), (2, 30, 301, 300)])
def test_float64_obs_is_written_by_the_expansion_kernel_itself(native, shape):
    """d2d_set_obs_dtype(D2D_F64): D2D_BUF_OBS is float64 [B, N, 6N] with exactly the float32 values (obs_fn.py:51 builds
    float64 arrays); switching back and forth re-sizes the library's own block; a bound block that is too small is refused."""
    import torch
    b, rbs, cues, dues = shape
    n = cues + dues
    sim, pos, raw = _batch(b, rbs, cues, dues, rng_seed=sum(shape) + 4)
    h = sim.handle
    h.set_obs_mode(native.OBS_LINEAR)
    sim.step_arrays(raw)
    o32 = sim.fetch(native.BUF_OBS).copy()
    h.set_obs_dtype(native.F64)
    sim.step_arrays(raw)
    o64 = sim.fetch(native.BUF_OBS)
    assert o64.dtype == np.float64 and o64.shape == (b, n, 6 * n) and np.array_equal(o64, o32.astype(np.float64))
    small = torch.empty(b * n * 6 * n, dtype=torch.float32, device='cuda')
    h.bind_buffer(native.BUF_OBS, small.data_ptr(), small.numel() * 4)      # float32-sized: too small for float64
    with pytest.raises(native.NativeError):
        sim.step_arrays(raw)
    h.set_obs_dtype(native.F32)
    sim.step_arrays(raw)
    torch.cuda.synchronize()
    assert np.array_equal(small.cpu().numpy().reshape(b, n, 6 * n), o32)
    with pytest.raises(native.NativeError):
        h.set_obs_dtype(7)
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


def test_obs_block_placement_trials(native):
    """VecD2DEnv(placement_trials=K), OPT-IN (default 0: a plain constructor allocates nothing beyond its own buffers): the first
    reset() times K candidate obs blocks, keeps the fastest, never holds more than placement_budget_bytes of candidates +
    paddings, leaves the caller's caching allocator alone; the observation it returns, and every later step, are
    bit-identical to an env that took the first block."""
    import torch
    from gym_d2d_amd.envs import VecD2DEnv
    cfg = {'num_rbs': 25, 'num_cues': 25, 'num_due_pairs': 25}
    plain = VecD2DEnv(dict(cfg), num_envs=256, cue_actions='traffic')
    assert plain._placement_trials == 0                                        # the default
    mine = torch.empty(64 << 20, dtype=torch.uint8, device='cuda'); del mine   # a block of the CALLER's in torch's cache
    cached = torch.cuda.memory_reserved()
    tuned = VecD2DEnv(dict(cfg), num_envs=256, cue_actions='traffic', placement_trials=3)
    o0, o1 = plain.reset(seed=4), tuned.reset(seed=4)
    assert plain.placement is None
    assert len(tuned.placement['us_per_step']) == 3 and 0 <= tuned.placement['chosen'] < 3 and tuned.placement['buffer'] == 'obs'
    assert torch.cuda.memory_reserved() >= cached                              # no empty_cache() behind the caller's back
    assert torch.equal(o0, o1) and o1.data_ptr() == tuned._t['obs'].data_ptr()
    for k in range(3):
        act = torch.randint(0, 25 * 21, (256, 25), device=plain.device, dtype=torch.int32)
        a, b = plain.step(act), tuned.step(act)
        assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    o0, o1 = plain.reset(), tuned.reset()                                      # trials run once per env
    assert torch.equal(o0, o1) and len(tuned.placement['us_per_step']) == 3
    # the budget: a 15 MB obs block + paddings of 2 - 22 MB under a 40 MB cap leaves room for ONE more candidate at most
    capped = VecD2DEnv(dict(cfg), num_envs=256, cue_actions='traffic', placement_trials=8, placement_budget_bytes=40 << 20)
    capped.reset(seed=4)
    assert len(capped.placement['us_per_step']) <= 2 and capped.placement['transient_bytes'] <= 40 << 20
    # the compact-obs step: the table is the block that is placed
    from gym_d2d_amd.envs.obs_fn import OwnLinkObsFunction
    t0 = VecD2DEnv({'num_rbs': 64, 'num_cues': 128, 'num_due_pairs': 128, 'obs_fn': OwnLinkObsFunction}, num_envs=512)
    t1 = VecD2DEnv({'num_rbs': 64, 'num_cues': 128, 'num_due_pairs': 128, 'obs_fn': OwnLinkObsFunction}, num_envs=512, placement_trials=4)
    a, b = t0.reset(seed=2), t1.reset(seed=2)
    assert t1.placement['buffer'] == 'table' and 1 <= len(t1.placement['us_per_step']) <= 4 and torch.equal(a, b)
    act = torch.randint(0, 64 * 21, (512, 256), device=t0.device, dtype=torch.int32)
    ra, rb_ = t0.step(act), t1.step(act)
    assert torch.equal(ra[0], rb_[0]) and torch.equal(ra[1], rb_[1]) and rb_[0].data_ptr() == t1._t['table'].data_ptr()
    for e in (plain, tuned, capped, t0, t1):
        e.close()


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
    sim.set_positions(case.pos[None].astype(np.float32))
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
