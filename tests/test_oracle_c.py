"""The C restatement of the oracle (oracle/c/d2d_oracle.c) against the NumPy oracle - which tests/test_oracle_golden.py
pins to the reference's golden vectors - and directly against the goldens that use the log-distance model.  CPU only."""
import numpy as np
import pytest

from golden_util import GOLDEN_DIR, rel_err
from oracle import c_oracle
from oracle import d2d_oracle as orc
from sim_util import random_layout


def _case(rng, envs, cues, dues, rbs, ple=2.0):
    ids, cfgs, is_bs = orc.device_configs(cues, dues)
    cols = orc.device_columns(cfgs, is_bs)
    pos = random_layout(rng, envs, cues, dues).astype(np.float64)
    tx = np.array([1 + c for c in range(cues)] + [1 + cues + 2 * p for p in range(dues)])
    rx = np.array([0] * cues + [2 + cues + 2 * p for p in range(dues)])
    ty = np.array([1] * cues + [3] * dues)
    lv = orc.pwr_levels_for(ty)
    act = rng.integers(0, rbs * lv[None, :], (envs, cues + dues))
    return pos, tx, rx, ty, act, cols, orc.PathLossSpec('log_distance', 2.1, ple=ple)


@pytest.mark.parametrize('shape', [(3, 4, 5, 2, 2.0), (2, 25, 25, 25, 2.0), (2, 40, 30, 3, 3.5), (1, 0, 6, 2, 2.0), (2, 7, 0, 3, 2.0)])
@pytest.mark.parametrize('min_cap', [0.0, 0.3])
def test_c_oracle_matches_numpy_oracle(shape, min_cap):
    envs, cues, dues, rbs, ple = shape
    rng = np.random.default_rng(envs * 1000 + cues * 10 + dues)
    pos, tx, rx, ty, act, cols, spec = _case(rng, envs, cues, dues, rbs, ple)
    act[0, 0] = -5                                                  # Python floor semantics for a negative action
    want = orc.full_step(pos, tx, rx, ty, act, cols, spec, min_capacity_mbps=min_cap)
    for threads in (1, 3):
        got = c_oracle.full_step(pos, tx, rx, ty, act, cols, spec, min_capacity_mbps=min_cap, threads=threads)
        assert np.array_equal(got['rb'], want['rb']) and np.array_equal(got['pwr'], want['pwr'])
        for k in ('sinr_db', 'snr_db', 'rate_bps', 'capacity_mbps', 'reward', 'table', 'obs'):
            assert rel_err(got[k], want[k]) <= 1e-12, (k, threads)


def test_c_oracle_zero_distance_is_the_reference_failure():
    """d = 0 -> log10(0): the reference raises 'math domain error'; both oracles produce a non-finite SINR there."""
    rng = np.random.default_rng(1)
    pos, tx, rx, ty, act, cols, spec = _case(rng, 1, 2, 2, 2)
    pos[0, 1] = 0.0                                                 # cue00 on the base station
    got = c_oracle.full_step(pos, tx, rx, ty, act, cols, spec)
    assert not np.isfinite(got['sinr_db'][0, 0])


@pytest.mark.parametrize('name', ['case01_default', 'case02_collisions', 'case03_stress256', 'case04_one_rb', 'case05_due_subset',
                                  'case06_downlink', 'case07_device_config', 'case09_ple35', 'case11_min_capacity'])
def test_c_oracle_against_reference_goldens(name):
    """Golden cases recorded from the imported reference that run a log-distance model: the C oracle lands on the
    reference's own numbers (not only on the NumPy restatement of them) - decode, SINR, SNR, rate, capacity, the
    SystemCapacity reward (with the recorded min-capacity variants) and the LinearObs rows the fixture holds."""
    from golden_util import load_case
    case = load_case(name)
    pl = case.meta['path_loss']
    assert pl['kind'] == 'log_distance'
    spec = orc.PathLossSpec('log_distance', case.meta['carrier_freq_GHz'], ple=pl['ple'])
    cols = orc.device_columns(case.cfgs, case.is_bs)
    pos = case.pos[None].astype(np.float64)
    npa = case.meta['num_pwr_actions']
    for k, st in enumerate(case.steps):
        n = len(st.keys)
        levels = np.array([{1: npa['cue'], 2: npa['mbs'], 3: npa['due']}[int(t)] for t in st.link_type])
        # the fixture holds decoded (rb, pwr); a raw action that decodes to them exercises the C decode as well
        act = (st.rb.astype(np.int64) * levels + st.pwr.astype(np.int64))[None]
        if (st.pwr >= levels).any() or (st.pwr < 0).any():
            continue                                            # array-form actions outside the alphabet: no raw equivalent
        mins = [0.0] + [float(f[len('reward_system_capacity_min'):].replace('p', '.')) for f in vars(st)
                        if f.startswith('reward_system_capacity_min')]
        for m in mins:
            got = c_oracle.full_step(pos, st.link_tx, st.link_rx, st.link_type, act, cols, spec, pwr_levels=levels,
                                     min_capacity_mbps=m)
            want_r = st.reward_system_capacity if m == 0.0 else getattr(st, 'reward_system_capacity_min' + str(m).replace('.', 'p'), None)
            if want_r is not None:
                assert rel_err(np.full(n, got['reward'][0]), want_r) <= 1e-12, (name, k, m)
        assert (got['rb'][0] == st.rb).all() and (got['pwr'][0] == st.pwr).all(), (name, k)
        for f in ('sinr_db', 'snr_db', 'rate_bps', 'capacity_mbps'):
            assert rel_err(got[f][0], getattr(st, f)) <= 1e-12, (name, k, f)
        assert rel_err(got['table'][0], st.obs_table) <= 1e-12
        assert rel_err(got['obs'][0, st.obs_rows], st.obs) <= 1e-12
