"""CPU oracle for the GymD2D per-step SINR / interference path.  TEST INFRASTRUCTURE ONLY.

This module is a NumPy float64 restatement of the reference algorithm
(davidcotton/gym-d2d @ v0.0.3).  It exists so the HIP kernels can be checked; it is
NOT part of the product.  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import it.  Nothing under ``gym_d2d_amd/``
imports it, and the product path raises if the HIP library is missing.

Parity status: PINNED.  ``tests/golden/*.npz`` hold inputs/outputs captured by running
the imported reference in the build container (``tests/golden/make_golden.py``);
``tests/test_oracle_golden.py`` checks every function below against them (<= 1e-12
relative) and against the known-answer values in the reference's own unit tests
(test_path_loss.py, test_conversion.py, test_device.py).

A second, independently written restatement in plain C lives in ``oracle/c/d2d_oracle.c`` (front end:
``oracle/c_oracle.py``; log-distance model only); ``tests/test_oracle_c.py`` holds it to this module and to the goldens.

Everything follows the reference's dB-domain arithmetic literally, in float64, batched
over a leading env axis B.  Citations are ``file:line`` under /root/reference/src/gym_d2d.

Shapes: B envs, D devices per env (0 = base station, 1..C = CUEs, then DUE tx/rx
interleaved - devices.py:20-25), N links per step.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict, Optional, Sequence

import numpy as np

SPEED_OF_LIGHT = 299792458.0          # path_loss.py:9
UPLINK, DOWNLINK, SIDELINK = 1, 2, 3  # link_type.py:4-7

# device.py:12-41 - default link-budget tables.
_BASE = {'num_PRB': 1, 'num_subcarriers': 12, 'subcarrier_spacing_kHz': 15.0}
DEFAULT_BS = dict(_BASE, max_tx_power_dBm=46.0, antenna_height_m=23.0, tx_antenna_gain_dBi=17.5,
                  rx_antenna_gain_dBi=17.5, thermal_noise_dBm=-118.4, noise_figure_dB=2.0, sinr_dB=-7.0,
                  ix_margin_dB=2.0, cable_loss_dB=2.0, masthead_amplifier_gain_dB=2.0)
DEFAULT_UE = dict(_BASE, max_tx_power_dBm=23.0, antenna_height_m=1.5, tx_antenna_gain_dBi=0.0,
                  rx_antenna_gain_dBi=0.0, thermal_noise_dBm=-104.5, noise_figure_dB=7.0, sinr_dB=-10.0,
                  ix_margin_dB=3.0, control_channel_overhead_dB=1.0, body_loss_dB=3.0)


# --------------------------------------------------------------------------- conversions
def db_to_linear(db):
    """conversion.py:4-13  pow(10, dB/10)."""
    return np.power(10.0, np.asarray(db, dtype=np.float64) / 10.0)


def linear_to_db(lin):
    """conversion.py:16-25  10*log10(x)."""
    return 10.0 * np.log10(np.asarray(lin, dtype=np.float64))


def dbm_to_w(dbm):
    """conversion.py:28-29."""
    return db_to_linear(dbm) / 1000.0


def w_to_dbm(w):
    """conversion.py:32-33."""
    return linear_to_db(np.asarray(w, dtype=np.float64) * 1000.0)


# --------------------------------------------------------------------------- device columns
@dataclass
class DeviceColumns:
    """Per-device derived link-budget columns, each shape [D] float64."""
    eirp_off_db: np.ndarray    # eirp_dBm(p) - p        device.py:51-60,134-135,158-159
    rx_off_db: np.ndarray      # rx_signal_level(e, pl) - (e - pl)   device.py:62-72,137-140,161-162
    noise_dbm: np.ndarray      # thermal_noise_dBm      device.py:118-119
    sens_dbm: np.ndarray       # rx_sensitivity_dBm     device.py:74-80
    bw_hz: np.ndarray          # rb_bandwidth_kHz*1000  device.py:85-95, simulator.py:150
    ant_h_m: np.ndarray        # antenna_height_m       device.py:101-103
    tx_gain_dbi: np.ndarray    # raw columns, for custom path-loss plugins
    rx_gain_dbi: np.ndarray


def device_configs(num_cues: int, num_due_pairs: int, *, num_subcarriers=12, subcarrier_spacing_kHz=15,
                   cue_max_tx_power_dBm=23, due_max_tx_power_dBm=20,
                   overrides: Optional[Dict[str, dict]] = None):
    """Device ids (order of devices.py:20-25) and merged config dicts (simulator.py:18-50).

    ``overrides`` is the parsed device_config_file JSON ({id: {'position':..., 'config': {...}}}).
    A device present there takes its 'config' entry INSTEAD of the env-level defaults
    (simulator.py:31), then merged over the class defaults (device.py:132,156).
    """
    overrides = overrides or {}
    base = {'num_subcarriers': num_subcarriers, 'subcarrier_spacing_kHz': subcarrier_spacing_kHz}
    ids, cfgs, is_bs = [], [], []

    def pick(dev_id, default):
        return overrides.get(dev_id, {}).get('config', default)

    ids.append('mbs'); is_bs.append(True)
    cfgs.append({**DEFAULT_BS, **pick('mbs', base)})
    for i in range(num_cues):
        dev_id = f'cue{i:02d}'
        ids.append(dev_id); is_bs.append(False)
        cfgs.append({**DEFAULT_UE, **pick(dev_id, {**base, 'max_tx_power_dBm': cue_max_tx_power_dBm})})
    for i in range(2 * num_due_pairs):
        dev_id = f'due{i:02d}'
        ids.append(dev_id); is_bs.append(False)
        cfgs.append({**DEFAULT_UE, **pick(dev_id, {**base, 'max_tx_power_dBm': due_max_tx_power_dBm})})
    return ids, cfgs, np.asarray(is_bs)


def device_columns(cfgs: Sequence[dict], is_bs: np.ndarray) -> DeviceColumns:
    d = len(cfgs)
    col = {k: np.zeros(d) for k in ('eirp', 'rxo', 'noise', 'sens', 'bw', 'h', 'txg', 'rxg')}
    for k, (c, bs) in enumerate(zip(cfgs, is_bs)):
        eirp = c['tx_antenna_gain_dBi'] - c['ix_margin_dB']              # device.py:60
        rxo = c['rx_antenna_gain_dBi']                                    # device.py:72
        if bs:
            eirp = eirp - c['cable_loss_dB'] + c['masthead_amplifier_gain_dB']   # device.py:135
            rxo = rxo - c['cable_loss_dB'] + c['masthead_amplifier_gain_dB']     # device.py:138-140
        else:
            eirp = eirp - c['body_loss_dB']                               # device.py:159
            rxo = rxo - c['body_loss_dB']                                 # device.py:162
        col['eirp'][k] = eirp
        col['rxo'][k] = rxo
        col['noise'][k] = c['thermal_noise_dBm']
        col['sens'][k] = c['noise_figure_dB'] + c['thermal_noise_dBm'] + c['sinr_dB']   # device.py:74-80
        col['bw'][k] = int(c['num_subcarriers']) * int(c['subcarrier_spacing_kHz']) * 1000   # device.py:85-95
        col['h'][k] = c['antenna_height_m']
        col['txg'][k] = c['tx_antenna_gain_dBi']
        col['rxg'][k] = c['rx_antenna_gain_dBi']
    return DeviceColumns(col['eirp'], col['rxo'], col['noise'], col['sens'], col['bw'], col['h'],
                         col['txg'], col['rxg'])


# --------------------------------------------------------------------------- path loss
def pl_constant_db(carrier_freq_ghz: float, ple: float) -> float:
    """path_loss.py:28-39."""
    return 10 * ple * math.log10(carrier_freq_ghz * 1e9) + 10 * ple * math.log10((4 * math.pi) / SPEED_OF_LIGHT)


@dataclass
class PathLossSpec:
    """kind: 'log_distance' | 'cost_hata' | 'table'."""
    kind: str = 'log_distance'
    carrier_freq_ghz: float = 2.1
    ple: float = 2.0
    area: str = 'suburban'                  # cost_hata: 'urban' | 'suburban' | 'rural'
    table_db: Optional[np.ndarray] = None   # 'table': [D, D] or [B, D, D] dB, index [tx_dev, rx_dev]


def path_loss_db(spec: PathLossSpec, dist_m, h_tx=None, h_rx=None):
    """Path loss in dB for distance array ``dist_m`` (any shape; h_* broadcastable)."""
    dist_m = np.asarray(dist_m, dtype=np.float64)
    if spec.kind == 'log_distance':
        # path_loss.py:65-66
        with np.errstate(divide='ignore'):
            return 10 * spec.ple * np.log10(dist_m) + pl_constant_db(spec.carrier_freq_ghz, spec.ple)
    if spec.kind == 'cost_hata':
        # path_loss.py:95-123
        f = spec.carrier_freq_ghz * 1000
        d_km = dist_m / 1000
        if spec.area == 'urban':
            if f >= 200:
                a_hc = 8.29 * np.log10(1.54 * h_rx) ** 2 - 1.1
            else:
                a_hc = 3.2 * np.log10(11.75 * h_rx) ** 2 - 4.97
            c = 3
        else:
            a_hc = (1.1 * math.log10(f) - 0.7) * h_rx - (1.56 * math.log10(f) - 0.8)
            c = 0
        with np.errstate(divide='ignore'):
            return (46.3 + 33.9 * math.log10(f) - 13.82 * np.log10(h_tx) - a_hc
                    + (44.9 - 6.55 * np.log10(h_tx)) * np.log10(d_km) + c)
    raise ValueError(spec.kind)


def pair_path_loss_db(spec, pos, link_tx, link_rx, cols):
    """PL[b, j, i] = path loss from tx of link j to rx of link i.  pos [B, D, 2]."""
    if spec.kind == 'table':
        t = np.asarray(spec.table_db, dtype=np.float64)
        if t.ndim == 2:
            t = t[None]
        return t[:, link_tx[:, None], link_rx[None, :]]
    tx = pos[:, link_tx, :]                                        # [B, N, 2]
    rx = pos[:, link_rx, :]
    dx = tx[:, :, None, 0] - rx[:, None, :, 0]
    dy = tx[:, :, None, 1] - rx[:, None, :, 1]
    dist = (dx ** 2 + dy ** 2) ** 0.5                               # position.py:11-12
    return path_loss_db(spec, dist, cols.ant_h_m[link_tx][None, :, None], cols.ant_h_m[link_rx][None, None, :])


# --------------------------------------------------------------------------- action decode
def decode_actions(raw, pwr_levels):
    """d2d_env.py:93-96  rb = a // P ; pwr = a % P (Python floor semantics; min power NOT added back)."""
    raw = np.asarray(raw, dtype=np.int64)
    p = np.asarray(pwr_levels, dtype=np.int64)
    return raw // p, raw % p


def pwr_levels_for(link_type, *, due_min=0, due_max=20, cue_max=23, mbs_max=46):
    """d2d_env.py:31-35,80-91  number of power levels per link type."""
    link_type = np.asarray(link_type)
    out = np.empty(link_type.shape, dtype=np.int64)
    out[link_type == SIDELINK] = due_max - due_min + 1
    out[link_type == UPLINK] = cue_max + 1
    out[link_type == DOWNLINK] = mbs_max + 1
    return out


# --------------------------------------------------------------------------- the step
@dataclass
class ShadowSpec:
    """ShadowingPathLoss (path_loss.py:69-81) with the draws of csrc/d2d_step.hip: beyond d0 every path-loss
    evaluation adds chi_db * z, z = Box-Muller of Philox4x32-10(counter = (first_env + b, step, j | i << 16, kind),
    key = seed); kind 0 = the SINR's signal / interferer terms, kind 1 = the SNR's re-evaluation (simulator.py:114)."""
    d0_m: float = 100.0
    chi_db: float = 2.7
    seed: int = 0
    step: int = 0
    first_env: int = 0

    def normals(self, env, j, i, kind):
        w = philox4x32_10(np.asarray(env, dtype=np.uint64) + np.uint64(self.first_env), np.uint64(self.step),
                          np.asarray(j, dtype=np.uint64) | (np.asarray(i, dtype=np.uint64) << np.uint64(16)),
                          np.uint64(kind), self.seed & 0xFFFFFFFF, (self.seed >> 32) & 0xFFFFFFFF)
        u1 = ((w[0] >> np.uint32(8)).astype(np.float64) + 0.5) * 2.0 ** -24
        u2 = (w[1] >> np.uint32(8)).astype(np.float64) * 2.0 ** -24
        return np.sqrt(-2.0 * np.log(u1)) * np.cos(2.0 * np.pi * u2)


def step(pos, link_tx, link_rx, rb, pwr, cols: DeviceColumns, spec: PathLossSpec, chunk: int = 64,
         shadow: Optional[ShadowSpec] = None):
    """simulator.py:77-154.  pos [B, D, 2] f64; link_tx/link_rx [N] device indices; rb, pwr [B, N].

    Returns dict of float64 [B, N]: sinr_db, snr_db, rate_bps, capacity_mbps.
    """
    pos = np.asarray(pos, dtype=np.float64)
    link_tx = np.asarray(link_tx); link_rx = np.asarray(link_rx)
    rb = np.asarray(rb); pwr = np.asarray(pwr, dtype=np.float64)
    b, n = rb.shape
    out = {k: np.empty((b, n)) for k in ('sinr_db', 'snr_db', 'rate_bps', 'capacity_mbps')}
    eye = np.eye(n, dtype=bool)
    noise = cols.noise_dbm[link_rx]
    h_tx, h_rx = cols.ant_h_m[link_tx], cols.ant_h_m[link_rx]
    for s in range(0, b, chunk):
        e = min(b, s + chunk)
        p = pos[s:e]
        eirp = pwr[s:e] + cols.eirp_off_db[link_tx][None, :]                    # device.py:60
        tab = None
        if spec.kind == 'table':
            tab = np.asarray(spec.table_db, dtype=np.float64)
            tab = tab[s:e] if tab.ndim == 3 else tab[None]

        def pl_of(bi, j, i, kind=0):
            """PL from the tx of link j to the rx of link i in env bi (index arrays)."""
            if tab is not None:
                return tab[bi if tab.shape[0] > 1 else 0, link_tx[j], link_rx[i]]
            t = p[bi, link_tx[j]]; r = p[bi, link_rx[i]]
            dist = ((t[..., 0] - r[..., 0]) ** 2 + (t[..., 1] - r[..., 1]) ** 2) ** 0.5   # position.py:11-12
            pl = path_loss_db(spec, dist, h_tx[j], h_rx[i])
            if shadow is not None:                                              # path_loss.py:76-79
                pl = pl + np.where(dist > shadow.d0_m, shadow.chi_db * shadow.normals(bi + s, j, i, kind), 0.0)
            return pl

        bi, ii = np.meshgrid(np.arange(e - s), np.arange(n), indexing='ij')
        sig = eirp - pl_of(bi, ii, ii) + cols.rx_off_db[link_rx][None, :]       # simulator.py:93
        sig_snr = sig if shadow is None else eirp - pl_of(bi, ii, ii, 1) + cols.rx_off_db[link_rx][None, :]
        # only links that share an RB interfere (simulator.py:95): evaluate exactly those (j -> i) pairs, like
        # the reference's per-RB sets do, and sum them per receiver in ascending j
        same = (rb[s:e, :, None] == rb[s:e, None, :]) & ~eye[None]              # [b, j, i]
        pb, pj, pi = np.nonzero(same)
        ix_mw = db_to_linear(eirp[pb, pj] - pl_of(pb, pj, pi))                  # simulator.py:97-101
        sum_ix = np.bincount(pb * n + pi, weights=ix_mw, minlength=(e - s) * n).reshape(e - s, n)
        sinr = sig - linear_to_db(sum_ix + db_to_linear(noise)[None, :])        # simulator.py:106-107
        snr = sig_snr - noise[None, :]                                          # simulator.py:114-115
        ok = sinr > cols.sens_dbm[link_rx][None, :]                             # simulator.py:123,149
        shannon = np.log2(1 + db_to_linear(sinr))
        out['sinr_db'][s:e] = sinr
        out['snr_db'][s:e] = snr
        out['rate_bps'][s:e] = np.where(ok, shannon, 0.0)                       # simulator.py:124-126
        out['capacity_mbps'][s:e] = np.where(ok, 1e-6 * cols.bw_hz[link_tx][None, :] * shannon, 0.0)  # :150-153
    return out


# --------------------------------------------------------------------------- rewards
def reward_system_capacity(cap, rb, link_type, min_capacity_mbps=0.0):
    """reward_fn.py:27-44  -> [B] (the same scalar goes to every agent of an env)."""
    cap = np.asarray(cap); rb = np.asarray(rb); link_type = np.asarray(link_type)
    b, n = cap.shape
    if n == 0:
        raise ZeroDivisionError('division by zero')                             # reward_fn.py:42
    d2d = link_type == SIDELINK
    same = (rb[:, :, None] == rb[:, None, :]) & ~np.eye(n, dtype=bool)[None]
    bad = same & d2d[None, :, None] & (~d2d)[None, None, :] & (cap <= min_capacity_mbps)[:, None, :]
    violated = bad.any(axis=(1, 2))
    return np.where(violated, -1.0, cap.sum(axis=1) / n)


def reward_shannon(sinr_db, min_sinr=-70.0):
    """reward_fn.py:52-57 -> [B, N]."""
    sinr_db = np.asarray(sinr_db)
    return np.where(sinr_db >= min_sinr, np.log2(1 + db_to_linear(sinr_db)), -1.0)


def reward_cue_sinr_shannon(sinr_db, rb, link_type, sinr_threshold_db=0.0):
    """reward_fn.py:65-78 -> [B, N]."""
    sinr_db = np.asarray(sinr_db); rb = np.asarray(rb); link_type = np.asarray(link_type)
    n = sinr_db.shape[1]
    same = (rb[:, :, None] == rb[:, None, :]) & ~np.eye(n, dtype=bool)[None]    # [b, i, j]
    bad = same & (link_type != SIDELINK)[None, None, :] & (sinr_db < sinr_threshold_db)[:, None, :]
    return np.where(bad.any(axis=2), -1.0, np.log2(1 + db_to_linear(sinr_db)))


# --------------------------------------------------------------------------- observations
def obs_table(pos, link_tx, link_rx, sinr_db, snr_db):
    """obs_fn.py:55-61  T[b, i] = (tx_x, tx_y, rx_x, rx_y, sinr_db, snr_db)  -> [B, N, 6]."""
    pos = np.asarray(pos, dtype=np.float64)
    return np.concatenate([pos[:, link_tx, :], pos[:, link_rx, :],
                           np.asarray(sinr_db)[:, :, None], np.asarray(snr_db)[:, :, None]], axis=2)


def expand_obs(table):
    """obs_fn.py:43-53  obs[b, i] = concat(T[i], T[0..i-1], T[i+1..N-1])  -> [B, N, 6N]."""
    table = np.asarray(table)
    b, n, w = table.shape
    flat = table.reshape(b, n * w)
    out = np.empty((b, n, n * w), dtype=table.dtype)
    # slot k of agent i shows link: i (k = 0), k-1 (1 <= k <= i), k (k > i)
    out[:] = flat[:, None, :]                                   # k > i: unshifted
    shifted = (np.arange(1, n)[None, :] <= np.arange(n)[:, None]).repeat(w, axis=1)    # [n, (n-1)w]: 1 <= k <= i
    if n > 1:
        np.copyto(out[:, :, w:], flat[:, None, :-w], where=shifted[None])
    out[:, :, :w] = table                                       # k = 0: own link
    return out


def full_step(pos, link_tx, link_rx, link_type, raw_actions, cols, spec, *, pwr_levels=None,
              min_capacity_mbps=0.0, with_obs=True, chunk=64):
    """d2d_env.py:62-71 end to end on arrays: decode -> step -> reward -> obs."""
    if pwr_levels is None:
        pwr_levels = pwr_levels_for(link_type)
    rb, pwr = decode_actions(raw_actions, np.asarray(pwr_levels)[None, :])
    st = step(pos, link_tx, link_rx, rb, pwr, cols, spec, chunk=chunk)
    st['rb'], st['pwr'] = rb, pwr
    st['reward'] = reward_system_capacity(st['capacity_mbps'], rb, link_type, min_capacity_mbps)
    st['table'] = obs_table(pos, link_tx, link_rx, st['sinr_db'], st['snr_db'])
    if with_obs:
        st['obs'] = expand_obs(st['table'])
    return st


# --------------------------------------------------------------------------- reset sampler
def sample_positions_from_uniforms(u, num_cues, num_due_pairs, cell_radius_m=500.0, d2d_radius_m=20.0,
                                   fixed: Optional[Dict[int, tuple]] = None):
    """simulator.py:61-75 + position.py:18-45 driven by an explicit uniform stream.

    ``u``: [B, D, T, 2] uniforms in [0,1): try t of device d uses (u[...,t,0] -> theta, u[...,t,1] -> r).
    CUE / DUE-tx use try 0 only; DUE-rx takes the first try that lands inside the cell
    (rejection loop, position.py:39-44).  Returns (pos [B, D, 2] f64, tries_used [B, D] int).
    Raises if T tries are not enough for some device.
    """
    u = np.asarray(u, dtype=np.float64)
    b, d, t, _ = u.shape
    assert d == 1 + num_cues + 2 * num_due_pairs
    pos = np.zeros((b, d, 2))
    used = np.zeros((b, d), dtype=np.int64)
    theta = 2 * np.pi * u[..., 0]
    for k in range(1, 1 + num_cues):                                            # position.py:24-28
        r = cell_radius_m * np.sqrt(u[:, k, 0, 1])
        pos[:, k, 0] = r * np.cos(theta[:, k, 0]); pos[:, k, 1] = r * np.sin(theta[:, k, 0]); used[:, k] = 1
    for p in range(num_due_pairs):
        kt = 1 + num_cues + 2 * p
        r = cell_radius_m * np.sqrt(u[:, kt, 0, 1])
        pos[:, kt, 0] = r * np.cos(theta[:, kt, 0]); pos[:, kt, 1] = r * np.sin(theta[:, kt, 0]); used[:, kt] = 1
    if fixed:
        for k, xy in fixed.items():
            pos[:, k, :] = xy; used[:, k] = 0
    for p in range(num_due_pairs):
        kt = 1 + num_cues + 2 * p
        kr = kt + 1
        if fixed and kr in fixed:
            continue
        done = np.zeros(b, dtype=bool)
        for j in range(t):                                                      # position.py:39-44
            r = d2d_radius_m * np.sqrt(u[:, kr, j, 1])
            x = pos[:, kt, 0] + r * np.cos(theta[:, kr, j]); y = pos[:, kt, 1] + r * np.sin(theta[:, kr, j])
            acc = ~done & ~(x ** 2 + y ** 2 > cell_radius_m ** 2)
            pos[acc, kr, 0] = x[acc]; pos[acc, kr, 1] = y[acc]; used[acc, kr] = j + 1
            done |= acc
        if not done.all():
            raise RuntimeError('rejection sampler ran out of tries')
    return pos, used


# --------------------------------------------------------------------------- counter-based RNG (reset path)
_PHILOX_M0, _PHILOX_M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
_PHILOX_W0, _PHILOX_W1 = 0x9E3779B9, 0xBB67AE85
_MASK32 = np.uint64(0xFFFFFFFF)


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """Philox4x32-10 (Salmon, Moraes, Dror, Shaw: 'Parallel random numbers: as easy as 1, 2, 3', SC'11), the
    stream csrc/d2d_reset.hip uses.  Inputs broadcastable uint arrays; returns 4 uint32 arrays."""
    c = [np.asarray(v, dtype=np.uint64) & _MASK32 for v in (c0, c1, c2, c3)]
    c = list(np.broadcast_arrays(*c))
    k0 = int(k0) & 0xFFFFFFFF; k1 = int(k1) & 0xFFFFFFFF
    for _ in range(10):
        p0 = _PHILOX_M0 * c[0]; p1 = _PHILOX_M1 * c[2]
        hi0, lo0 = p0 >> np.uint64(32), p0 & _MASK32
        hi1, lo1 = p1 >> np.uint64(32), p1 & _MASK32
        c = [hi1 ^ c[1] ^ np.uint64(k0), lo1, hi0 ^ c[3] ^ np.uint64(k1), lo0]
        k0 = (k0 + _PHILOX_W0) & 0xFFFFFFFF; k1 = (k1 + _PHILOX_W1) & 0xFFFFFFFF
    return [v.astype(np.uint32) for v in c]


def reset_uniforms(seed, episode, num_envs, num_devices, tries, first_env=0):
    """The uniforms the device-side reset consumes, counter (first_env + b, d, t, episode), key = 64-bit seed:
    u[b, d, t, 0] -> theta = (word0 >> 8) * 2^-24 in [0, 1); u[b, d, t, 1] -> radius = ((word1 >> 8) + 0.5) * 2^-24
    in the OPEN interval (0, 1), so no device is ever placed at distance 0 from its anchor."""
    b = (np.arange(num_envs, dtype=np.uint64) + np.uint64(first_env))[:, None, None]
    d = np.arange(num_devices, dtype=np.uint64)[None, :, None]
    t = np.arange(tries, dtype=np.uint64)[None, None, :]
    w = philox4x32_10(b, d, t, np.uint64(episode & 0xFFFFFFFF), seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)
    scale = 2.0 ** -24
    return np.stack([(w[0] >> np.uint32(8)) * scale, ((w[1] >> np.uint32(8)) + 0.5) * scale], axis=-1)
