"""Pin the NumPy oracle against the golden vectors captured from the reference (CPU only)."""
import math

import numpy as np
import pytest

from golden_util import case_names, known_answers, load_case, rel_err
from oracle import d2d_oracle as orc

TOL = 1e-12


def spec_for(case):
    pl = case.meta['path_loss']
    f = case.meta['carrier_freq_GHz']
    if pl['kind'] == 'log_distance':
        return orc.PathLossSpec('log_distance', f, ple=pl['ple'])
    if pl['kind'] == 'cost_hata':
        return orc.PathLossSpec('cost_hata', f, area=pl['area'])
    if pl['kind'] == 'custom_foo':
        # the fixture's user-defined model: 20 log10(d) - tx_gain - rx_gain, as a [D, D] table
        cols = orc.device_columns(case.cfgs, case.is_bs)
        d = np.hypot(case.pos[:, None, 0] - case.pos[None, :, 0], case.pos[:, None, 1] - case.pos[None, :, 1])
        with np.errstate(divide='ignore'):
            table = 20 * np.log10(d) - cols.tx_gain_dbi[:, None] - cols.rx_gain_dbi[None, :]
        return orc.PathLossSpec('table', f, table_db=table)
    raise AssertionError(pl)


def pwr_levels(case, link_type):
    m = case.meta
    return orc.pwr_levels_for(link_type, due_min=m['due_min_tx_power_dBm'], due_max=m['due_max_tx_power_dBm'],
                              cue_max=m['cue_max_tx_power_dBm'], mbs_max=m['mbs_max_tx_power_dBm'])


def shadow_statistics(sample):
    """per-link mean/std of sinr, snr and std of (sinr - snr) over the first axis of `sample` = (sinr, snr)."""
    sinr, snr = sample
    return {'sinr_mean': sinr.mean(0), 'sinr_std': sinr.std(0), 'snr_mean': snr.mean(0), 'snr_std': snr.std(0),
            'diff_std': (sinr - snr).std(0)}


def check_shadow_statistics(got, s, reps_got, slack=1e-9):
    """Compare per-link moments with the reference's captured ones (case13).  Standard errors: mean ~ std/sqrt(n),
    std ~ std/sqrt(2n); both samples are finite, bounds are 6 sigma of the combined error."""
    reps_ref = int(s.reps)
    for key in ('sinr', 'snr'):
        sd = np.maximum(getattr(s, f'{key}_std'), 1e-9)
        se_mean = sd * np.sqrt(1.0 / reps_ref + 1.0 / reps_got)
        assert (np.abs(got[f'{key}_mean'] - getattr(s, f'{key}_mean')) <= 6 * se_mean + slack).all(), key
        se_std = sd * np.sqrt(0.5 / reps_ref + 0.5 / reps_got) * 1.5      # x1.5: sums of log-normals are heavy tailed
        assert (np.abs(got[f'{key}_std'] - getattr(s, f'{key}_std')) <= 6 * se_std + slack).all(), key
    sd = np.maximum(s.diff_std, 1e-9)
    assert (np.abs(got['diff_std'] - s.diff_std) <= 6 * 1.5 * sd * np.sqrt(0.5 / reps_ref + 0.5 / reps_got) + slack).all()


def test_oracle_shadowing_matches_reference_distribution():
    """ShadowingPathLoss is stochastic: pin the oracle's draw structure (independent Gaussians for the SINR signal,
    every interferer, and the SNR re-evaluation; none within d0) through per-link moments."""
    case = load_case('case13_shadowing')
    s = case.steps[0]
    pl = case.meta['path_loss']
    assert pl['kind'] == 'shadowing'
    cols = orc.device_columns(case.cfgs, case.is_bs)
    spec = orc.PathLossSpec('log_distance', case.meta['carrier_freq_GHz'], ple=pl['ple'])
    reps = 1500
    sinr = np.empty((reps, len(s.keys))); snr = np.empty_like(sinr)
    for k in range(reps):
        sh = orc.ShadowSpec(pl['d0_m'], pl['chi_dB'], seed=99, step=k)
        out = orc.step(case.pos[None], s.link_tx, s.link_rx, s.rb[None], s.pwr[None], cols, spec, shadow=sh)
        sinr[k], snr[k] = out['sinr_db'][0], out['snr_db'][0]
    check_shadow_statistics(shadow_statistics((sinr, snr)), s, reps)
    # structure: links shorter than d0 (every DUE pair: <= 20 m) have a deterministic SNR
    d = np.hypot(*(case.pos[s.link_tx] - case.pos[s.link_rx]).T)
    assert (s.snr_std[d <= pl['d0_m']] < 1e-9).all() and (snr.std(0)[d <= pl['d0_m']] < 1e-9).all()
    assert (s.snr_std[d > pl['d0_m']] > 2.0).all()
    # chi = 0 degenerates to plain log-distance
    det = orc.step(case.pos[None], s.link_tx, s.link_rx, s.rb[None], s.pwr[None], cols, spec)
    zero = orc.step(case.pos[None], s.link_tx, s.link_rx, s.rb[None], s.pwr[None], cols, spec,
                    shadow=orc.ShadowSpec(pl['d0_m'], 0.0, seed=1, step=0))
    assert rel_err(zero['sinr_db'], det['sinr_db']) < TOL


@pytest.mark.parametrize('name', [n for n in case_names() if 'shadowing' not in n])
def test_oracle_matches_reference(name):
    case = load_case(name)
    cols = orc.device_columns(case.cfgs, case.is_bs)
    spec = spec_for(case)
    pos = case.pos[None]
    for k, s in enumerate(case.steps):
        n = len(s.keys)
        # ---- decode (d2d_env.py:93-101)
        if hasattr(s, 'raw') and (s.raw >= 0).all():
            rb, pwr = orc.decode_actions(s.raw[None], pwr_levels(case, s.link_type)[None])
            assert (rb[0] == s.rb).all() and (pwr[0] == s.pwr).all(), (name, k)
        if hasattr(s, 'raw_rb_pwr'):
            assert (s.raw_rb_pwr[:, 0] == s.rb).all() and (s.raw_rb_pwr[:, 1] == s.pwr).all()
        # ---- SINR / SNR / rate / capacity (simulator.py:77-154)
        out = orc.step(pos, s.link_tx, s.link_rx, s.rb[None], s.pwr[None], cols, spec)
        for f in ('sinr_db', 'snr_db', 'rate_bps', 'capacity_mbps'):
            assert rel_err(out[f][0], getattr(s, f)) < TOL, (name, k, f)
        # ---- rewards (reward_fn.py)
        r = orc.reward_system_capacity(out['capacity_mbps'], s.rb[None], s.link_type)
        assert rel_err(np.full(n, r[0]), s.reward_system_capacity) < TOL
        assert rel_err(orc.reward_shannon(out['sinr_db'])[0], s.reward_shannon) < TOL
        assert rel_err(orc.reward_cue_sinr_shannon(out['sinr_db'], s.rb[None], s.link_type)[0],
                       s.reward_cue_sinr_shannon) < TOL
        for f in vars(s):
            if f.startswith('reward_system_capacity_min'):
                m = float(f[len('reward_system_capacity_min'):].replace('p', '.'))
                r = orc.reward_system_capacity(out['capacity_mbps'], s.rb[None], s.link_type, m)
                assert rel_err(np.full(n, r[0]), getattr(s, f)) < TOL, (name, k, f)
        if hasattr(s, 'reward_env'):
            assert rel_err(s.reward_env, s.reward_system_capacity) == 0.0
        # ---- observations (obs_fn.py:43-61)
        table = orc.obs_table(pos, s.link_tx, s.link_rx, out['sinr_db'], out['snr_db'])
        assert rel_err(table[0], s.obs_table) < TOL
        full = orc.expand_obs(table)
        assert full.shape == (1, n, 6 * n)
        assert rel_err(full[0, s.obs_rows], s.obs) < TOL
        # positions inside obs are copies: exact
        assert (full[0, s.obs_rows][:, :4] == s.obs[:, :4]).all()


def test_minus_one_branch_is_exercised():
    """case11 must actually contain both outcomes of reward_fn.py:38-42."""
    case = load_case('case11_min_capacity')
    seen = set()
    for s in case.steps:
        for f in vars(s):
            if f.startswith('reward_system_capacity_min'):
                seen.add(float(getattr(s, f)[0]) == -1.0)
    assert seen == {True, False}


def test_game_over_flips_on_step_10():
    case = load_case('case01_default')
    flags = [bool(s.game_over) for s in case.steps[1:]]
    assert flags == [False] * 9 + [True]                                        # d2d_env.py:16,68


def test_full_step_wrapper_matches():
    case = load_case('case02_collisions')
    s = case.steps[2]
    cols = orc.device_columns(case.cfgs, case.is_bs)
    st = orc.full_step(case.pos[None], s.link_tx, s.link_rx, s.link_type, s.raw[None], cols, spec_for(case),
                       pwr_levels=pwr_levels(case, s.link_type))
    assert rel_err(st['sinr_db'][0], s.sinr_db) < TOL
    assert rel_err(st['obs'][0, s.obs_rows], s.obs) < TOL
    assert rel_err(st['reward'], s.reward_env[:1]) < TOL


def test_device_configs_match_reference_merge():
    """simulator.py:18-50 + device.py defaults, incl. the device_config_file override route."""
    import json
    from golden_util import GOLDEN_DIR
    for name in ('case01_default', 'case07_device_config'):
        case = load_case(name)
        m = case.meta
        ov = json.loads((GOLDEN_DIR / 'case07_device_config.json').read_text()) if '07' in name else None
        ids, cfgs, is_bs = orc.device_configs(m['num_cues'], m['num_due_pairs'],
                                              num_subcarriers=m['num_subcarriers'],
                                              subcarrier_spacing_kHz=m['subcarrier_spacing_kHz'],
                                              cue_max_tx_power_dBm=m['cue_max_tx_power_dBm'],
                                              due_max_tx_power_dBm=m['due_max_tx_power_dBm'], overrides=ov)
        assert ids == case.ids
        assert (is_bs == case.is_bs).all()
        for a, b in zip(cfgs, case.cfgs):
            assert a == b


def test_known_answers_from_reference_unit_tests():
    kat = known_answers()
    ap = lambda a, b, tol=1e-5: abs(a - b) <= tol * max(1.0, abs(b))
    assert ap(orc.pl_constant_db(2.1, 2.0), kat['test_path_loss.py:11 pl_constant_dB(2.1,2.0)'], 1e-14)
    ld = orc.PathLossSpec('log_distance', 2.1, ple=2.0)
    assert ap(orc.path_loss_db(ld, 250.0), kat['test_path_loss.py:25 logdist 2.1GHz 250m'])
    assert ap(orc.path_loss_db(ld, 500.0), kat['test_path_loss.py:27 logdist 2.1GHz 500m'])
    hu = orc.PathLossSpec('cost_hata', 2.1, area='urban')
    assert ap(orc.path_loss_db(hu, 250.0, 23.0, 1.5), kat['test_path_loss.py:48 hata urban bs->ue 250m'], 1e-12)
    assert ap(orc.path_loss_db(hu, 250.0, 1.5, 23.0), kat['test_path_loss.py:49 hata urban ue->bs 250m'], 1e-12)
    assert ap(orc.path_loss_db(hu, 500.0, 23.0, 1.5), kat['test_path_loss.py:51 hata urban bs->ue 500m'], 1e-12)
    assert ap(orc.path_loss_db(hu, 500.0, 1.5, 23.0), kat['test_path_loss.py:52 hata urban ue->bs 500m'], 1e-12)
    assert ap(orc.db_to_linear(1), kat['test_conversion.py:8 dB_to_linear(1)'])
    assert ap(orc.linear_to_db(2), kat['test_conversion.py:17 linear_to_dB(2)'])
    assert ap(orc.dbm_to_w(30), kat['test_conversion.py:31 dBm_to_W(30)'])
    assert ap(orc.w_to_dbm(0.2), kat['test_conversion.py:37 W_to_dBm(0.2)'])
    ids, cfgs, is_bs = orc.device_configs(1, 0)
    cols = orc.device_columns(cfgs, is_bs)
    assert 12 + cols.eirp_off_db[1] == kat['test_device.py:73-77 ue eirp(12)']
    assert 46 + cols.eirp_off_db[0] == kat['test_device.py:80-85 bs eirp(46)']
    assert cols.sens_dbm[1] == -107.5 and cols.sens_dbm[0] == pytest.approx(-123.4)


def test_expand_obs_layout_small():
    t = np.arange(3 * 6, dtype=np.float64).reshape(1, 3, 6)
    o = orc.expand_obs(t)[0]
    assert list(o[0]) == list(range(18))
    assert list(o[1]) == list(range(6, 12)) + list(range(0, 6)) + list(range(12, 18))
    assert list(o[2]) == list(range(12, 18)) + list(range(0, 12))


def test_sampler_properties():
    """position.py samplers: inside the cell; DUE rx within d2d radius (test_position.py:30-44)."""
    rng = np.random.default_rng(0)
    u = rng.random((64, 1 + 5 + 10, 16, 2))
    pos, used = orc.sample_positions_from_uniforms(u, 5, 5, 500.0, 20.0)
    assert (np.hypot(pos[..., 0], pos[..., 1]) <= 500.0 + 1e-9).all()
    tx = pos[:, 6::2]; rx = pos[:, 7::2]
    assert (np.hypot(*(tx - rx).transpose(2, 0, 1)) <= 20.0 + 1e-9).all()
    assert (pos[:, 0] == 0).all()


def test_oracle_sampler_matches_the_reference_samplers_on_the_same_uniforms():
    """Golden sampler_case15: the reference's get_random_position / get_random_position_nearby, as driven by
    Simulator.reset (position.py:18-45, simulator.py:61-75), fed with the Philox uniforms of the device-side reset.  The
    oracle's sampler must land on the same positions (same arithmetic, same consumption order, same rejection
    decisions), and reset_uniforms must regenerate exactly the stream that was fed."""
    import json
    from golden_util import GOLDEN_DIR
    z = np.load(GOLDEN_DIR / 'sampler_case15.npz')
    meta = json.loads(bytes(z['meta_json']).decode())
    assert [c['tag'] for c in meta['configs']] == ['default_radii', 'tight_cell']
    for c in meta['configs']:
        u, ref = z[c['tag'] + '_u'], z[c['tag'] + '_pos']
        d = 1 + c['num_cues'] + 2 * c['num_due_pairs']
        again = orc.reset_uniforms(c['seed'], c['episode'], c['num_envs'], d, u.shape[2], first_env=c['first_env'])
        assert np.array_equal(again, u)
        assert u[..., 1].min() > 0.0 and u.max() < 1.0            # radius draws on the open interval: never distance 0
        pos, used = orc.sample_positions_from_uniforms(u, c['num_cues'], c['num_due_pairs'], c['cell_radius_m'],
                                                       c['d2d_radius_m'])
        assert np.abs(pos - ref).max() <= 1e-12 * c['cell_radius_m'], c['tag']
        assert used.max() == c['max_tries_used'], c['tag']        # the rejection loop stopped at the same try everywhere
    assert meta['configs'][1]['max_tries_used'] >= 5               # the tight cell really exercises rejection
