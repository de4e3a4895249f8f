"""GPU tests of the float64-position route (d2d_set_positions_f64, kernels compiled with OPT_XPOS): the reference works on Python
floats throughout (position.py:7-12,18-45; simulator.py:61-75), and float32 ABSOLUTE coordinates carry 3e-5 m at 500 m - a
receiver 0.1 m from a transmitter is then off by 3e-4 relative, 2e-5 on sinr_db, from input rounding alone.  The golden side
(tests/golden/case16_unrounded_*) runs in test_gpu_parity.py; here: random float64 layouts against the oracle at the bar, the
gap of the float32 upload beside it, bit identity between the kernels that serve the route, and its on / off rules."""
import numpy as np
import pytest

from golden_util import rel_err
from oracle import d2d_oracle as orc
from sim_util import default_links, search_variants, snapshot, assert_same

pytestmark = pytest.mark.gpu
TOL = 1e-5
FIELDS = (('sinr_db', 'BUF_SINR_DB'), ('snr_db', 'BUF_SNR_DB'), ('rate_bps', 'BUF_RATE_BPS'), ('capacity_mbps', 'BUF_CAPACITY'))


def _layout64(rng, num_envs, cues, dues, cell_radius=500.0, d2d_radius=20.0):
    """float64 positions [B, D, 2] exactly as the reference's samplers produce them (oracle restatement of position.py:18-45)."""
    d = 1 + cues + 2 * dues
    pos, _ = orc.sample_positions_from_uniforms(rng.random((num_envs, d, 32, 2)), cues, dues, cell_radius, d2d_radius)
    assert pos.dtype == np.float64 and (pos != pos.astype(np.float32)).any()
    return pos


def _sim(num_envs, rbs, cues, dues, **cfg):
    from gym_d2d_amd.simulator import Simulator
    sim = Simulator(dict(num_rbs=rbs, num_cues=cues, num_due_pairs=dues, num_envs=num_envs, **cfg), max_links=cues + dues)
    sim.set_links(sim.default_link_keys())
    return sim


def _raw(rng, sim, num_envs, rbs, cues, dues):
    p = sim.config.num_pwr_actions
    return np.concatenate([rng.integers(0, rbs * p['cue'], (num_envs, cues)), rng.integers(0, rbs * p['due'], (num_envs, dues))],
                          axis=1).astype(np.int32)


def _oracle(sim, pos, raw, spec=None, **kw):
    ids, cfgs, is_bs = orc.device_configs(sim.config.num_cues, sim.config.num_due_pairs)
    cols = orc.device_columns(cfgs, is_bs)
    tx, rx, ty = default_links(sim.config.num_cues, sim.config.num_due_pairs)
    return orc.full_step(pos, tx, rx, ty, raw, cols, spec or orc.PathLossSpec(), chunk=16, **kw)


def _errors(sim, native, ref):
    return {f: rel_err(sim.fetch(getattr(native, buf)), ref[f]) for f, buf in FIELDS}


@pytest.mark.parametrize('shape', [(256, 25, 25, 25), (16, 256, 256, 256), (64, 8, 30, 70)], ids=['256x50', '16x512', '64x100_padded'])
def test_float64_layouts_hold_the_bar_where_float32_uploads_do_not(native, shape):
    """The oracle on the float64 layout is the reference's answer for the reference's inputs.  Through d2d_set_positions_f64 the
    HIP path holds 1e-5 with the margin it has on float32-representable inputs (a few 1e-7 .. 2e-6); the same layout uploaded
    as float32 is further off than that on every field, and beyond the bar wherever a pair a few metres apart sits far out."""
    b, rbs, cues, dues = shape
    rng = np.random.default_rng(2026 + b)
    sim = _sim(b, rbs, cues, dues)
    pos = _layout64(rng, b, cues, dues)
    raw = _raw(rng, sim, b, rbs, cues, dues)
    ref = _oracle(sim, pos, raw, with_obs=True)
    sim.handle.set_obs_mode(native.OBS_LINEAR)
    sim.set_positions(pos)                                   # float64 array -> (hi, lo) pairs
    sim.step_arrays(raw)
    assert sim.check_flags() == 0
    exact = _errors(sim, native, ref)
    assert max(exact.values()) <= TOL, exact
    assert max(exact.values()) <= 3e-6, exact                # the arithmetic margin, not the input rounding
    assert rel_err(sim.fetch(native.BUF_REWARD)[:, 0], ref['reward']) <= TOL
    table = sim.fetch(native.BUF_OBS_TABLE)
    assert (table[..., :4] == ref['table'][..., :4].astype(np.float32)).all(), 'positions in the table: the float32 rounding of the float64 value'
    assert rel_err(sim.fetch(native.BUF_OBS), ref['obs']) <= TOL
    sim.set_positions(pos.astype(np.float32))                # what every round before the sixth did with such a layout
    sim.step_arrays(raw)
    rounded = _errors(sim, native, ref)
    assert rounded['sinr_db'] > 2.0 * exact['sinr_db'], (exact, rounded)
    print(f'\n{shape}: float64 upload {exact}; float32 upload {rounded}')
    sim.handle.close()


@pytest.mark.parametrize('model', ['ple37', 'hata'])
def test_float64_layouts_power_law_kernels(native, model):
    from gym_d2d_amd.path_loss import AreaType, CostHataPathLoss, LogDistancePathLoss

    class Urban(CostHataPathLoss):
        def __init__(self, f):
            super().__init__(f, AreaType.URBAN)

    class Ple(LogDistancePathLoss):
        def __init__(self, f):
            super().__init__(f, ple=3.7)
    rng = np.random.default_rng(77)
    b, rbs, cues, dues = 32, 64, 64, 64
    sim = _sim(b, rbs, cues, dues, path_loss_model=Ple if model == 'ple37' else Urban)
    pos = _layout64(rng, b, cues, dues)
    raw = _raw(rng, sim, b, rbs, cues, dues)
    spec = orc.PathLossSpec('log_distance', 2.1, ple=3.7) if model == 'ple37' else orc.PathLossSpec('cost_hata', 2.1, area='urban')
    ref = _oracle(sim, pos, raw, spec)
    sim.set_positions(pos)
    for obs_mode in (native.OBS_TABLE, native.OBS_NONE):     # generic kernel with masks / the rollout kernel
        sim.handle.set_obs_mode(obs_mode)
        sim.step_arrays(raw)
        err = _errors(sim, native, ref)
        assert max(err.values()) <= TOL, (model, obs_mode, err)
    sim.handle.close()


def test_exact_position_kernels_are_bit_identical_and_fall_back_to_float32_bits(native):
    """Every kernel forms (tx_hi - rx_hi) + (tx_lo - rx_lo) in the same order: the mask walk, the member lists (= the rollout
    kernel here), the all-pairs sweep and the padded rollout kernel agree bit for bit on a float64 layout; a float64 ARRAY whose
    values float32 can hold runs the float32 kernels and gives their bits."""
    rng = np.random.default_rng(5)
    for b, rbs, cues, dues in ((24, 32, 64, 64), (24, 40, 50, 50)):          # a multiple of 64 links; a padded count
        sim = _sim(b, rbs, cues, dues)
        pos = _layout64(rng, b, cues, dues)
        raw = _raw(rng, sim, b, rbs, cues, dues)
        h = sim.handle
        h.set_obs_mode(native.OBS_TABLE)
        sim.set_positions(pos)
        for rid, param in ((native.REWARD_SYSTEM_CAPACITY, 0.0), (native.REWARD_SHANNON, -70.0), (native.REWARD_CUE_SINR_SHANNON, 0.0)):
            h.set_reward(rid, param)

            def run():
                sim.step_arrays(raw)
                return snapshot(sim, native)
            assert_same(search_variants(native, h, run))
        h.set_reward(native.REWARD_SYSTEM_CAPACITY, 0.0)
        f32 = pos.astype(np.float32)
        sim.set_positions(f32)
        sim.step_arrays(raw)
        a = snapshot(sim, native)
        sim.set_positions(f32.astype(np.float64))            # float64 dtype, float32 values: no low parts, same kernels
        sim.step_arrays(raw)
        c = snapshot(sim, native)
        for k in a:
            assert np.array_equal(a[k], c[k], equal_nan=True), k
        h.close()


def test_low_parts_follow_the_ways_positions_are_written(native):
    """d2d_set_positions on a sub-range keeps the other envs' low parts; a device-side reset or a float32 upload of all envs
    returns the handle to float32 coordinates."""
    rng = np.random.default_rng(9)
    b, rbs, cues, dues = 8, 16, 32, 32
    sim = _sim(b, rbs, cues, dues)
    pos = _layout64(rng, b, cues, dues)
    raw = _raw(rng, sim, b, rbs, cues, dues)
    ref = _oracle(sim, pos, raw)
    sim.set_positions(pos)
    sim.step_arrays(raw)
    full = sim.fetch(native.BUF_SINR_DB).copy()
    assert rel_err(full, ref['sinr_db']) <= 3e-6
    f32 = pos.astype(np.float32)
    sim.handle.set_positions(f32[2:4, :, 0], f32[2:4, :, 1], env_begin=2)     # envs 2, 3 become float32; the others keep (hi, lo)
    sim.step_arrays(raw)
    part = sim.fetch(native.BUF_SINR_DB).copy()
    keep = np.r_[0:2, 4:b]
    assert np.array_equal(part[keep], full[keep])
    sim.set_positions(f32)
    sim.step_arrays(raw)
    rounded = sim.fetch(native.BUF_SINR_DB).copy()
    assert np.array_equal(part[2:4], rounded[2:4]) and not np.array_equal(rounded[keep], full[keep])
    sim.handle.set_positions(pos[5:6, :, 0], pos[5:6, :, 1], env_begin=5)     # one env back to float64
    sim.step_arrays(raw)
    one = sim.fetch(native.BUF_SINR_DB).copy()
    assert np.array_equal(one[5], full[5]) and np.array_equal(one[np.r_[0:5, 6:b]], rounded[np.r_[0:5, 6:b]])
    sim.reset_device(seed=3)                                  # the sampler draws float32 coordinates
    sim.step_arrays(raw)
    after = sim.fetch(native.BUF_SINR_DB).copy()
    sim.set_positions(sim.positions())                       # ... so re-uploading what it drew changes nothing
    sim.step_arrays(raw)
    assert np.array_equal(after, sim.fetch(native.BUF_SINR_DB))
    sim.handle.close()


def test_batched_env_with_a_reference_saved_device_config_takes_the_file_precision(native, tmp_path):
    """A device_config_file the reference saved holds float64 coordinates (d2d_env.py:124-134); VecD2DEnv pins them in every env
    (simulator.py:65-66) and the step sees them exactly."""
    import json
    from golden_util import GOLDEN_DIR, load_case
    from gym_d2d_amd.envs import VecD2DEnv
    case = load_case('case16_unrounded_device_config')
    pinned = json.loads((GOLDEN_DIR / 'case16_unrounded_device_config.json').read_text())
    env = VecD2DEnv({'num_rbs': 8, 'num_cues': 6, 'num_due_pairs': 6, 'device_config_file': GOLDEN_DIR / 'case16_unrounded_device_config.json'},
                    num_envs=4, use_torch=False)
    env.reset(seed=11)
    sim = env.simulator
    pos = sim.positions()
    for k, dev_id in enumerate(case.ids):
        if dev_id in pinned and dev_id != 'mbs':
            assert (pos[:, k] == np.float32(pinned[dev_id]['position'])).all()
    assert sim.handle.get_buffer(native.BUF_LINK_POS)[0]      # rows refreshed
    # the pinned devices' float64 values reached the kernels: env 0 against the oracle on (sampled float32 + pinned float64)
    exact = pos.astype(np.float64)
    for k, dev_id in enumerate(case.ids):
        if dev_id in pinned and dev_id != 'mbs':
            exact[:, k] = pinned[dev_id]['position']
    rng = np.random.default_rng(1)
    raw = _raw(rng, sim, 4, 8, 6, 6)
    env.step(raw)
    cols = orc.device_columns(case.cfgs, case.is_bs)
    tx, rx, ty = default_links(6, 6)
    ref = orc.full_step(exact, tx, rx, ty, raw, cols, orc.PathLossSpec())
    err = _errors(sim, native, ref)
    assert max(err.values()) <= 3e-6, err
    env.close()
