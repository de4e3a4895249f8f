"""ctypes front end of the C restatement of the oracle (oracle/c/d2d_oracle.c).  TEST INFRASTRUCTURE ONLY - the same
standing as d2d_oracle.py: only tests/, ``__graft_entry__`` and ``bench.py``'s cpu_baseline leg may import it.

    full_step(pos, link_tx, link_rx, link_type, raw_actions, cols, spec, ...) -> dict   (same keys as d2d_oracle.full_step)

Log-distance path loss only (the default model); float64 throughout; ``threads`` > 1 runs envs in parallel (OpenMP).
"""
from __future__ import annotations

import ctypes as C
import subprocess
from pathlib import Path

import numpy as np

from . import d2d_oracle as orc

HERE = Path(__file__).resolve().parent / 'c'
LIB_PATH = HERE / 'libd2d_oracle_c.so'
_lib = None

_DP, _IP, _LP = C.POINTER(C.c_double), C.POINTER(C.c_int32), C.POINTER(C.c_int64)


class _Args(C.Structure):
    _fields_ = [('B', C.c_int32), ('D', C.c_int32), ('N', C.c_int32), ('R', C.c_int32),
                ('pos', _DP), ('link_tx', _IP), ('link_rx', _IP), ('link_type', _IP), ('pwr_levels', _IP), ('actions', _LP),
                ('eirp_off_db', _DP), ('rx_off_db', _DP), ('noise_dbm', _DP), ('sens_dbm', _DP), ('bw_hz', _DP),
                ('ple', C.c_double), ('pl_const_db', C.c_double), ('min_capacity_mbps', C.c_double),
                ('rb', _LP), ('pwr', _LP), ('sinr_db', _DP), ('snr_db', _DP), ('rate_bps', _DP), ('capacity_mbps', _DP),
                ('reward', _DP), ('table', _DP), ('obs', _DP), ('threads', C.c_int32)]


def build(force: bool = False) -> Path:
    src = HERE / 'd2d_oracle.c'
    if force or not LIB_PATH.exists() or LIB_PATH.stat().st_mtime < src.stat().st_mtime:
        subprocess.run(['bash', str(HERE / 'build.sh')], check=True)
    return LIB_PATH


def load():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(str(LIB_PATH))
        _lib.d2d_oracle_step.argtypes = [C.POINTER(_Args)]
        _lib.d2d_oracle_step.restype = C.c_int
        _lib.d2d_oracle_max_threads.restype = C.c_int
    return _lib


def max_threads() -> int:
    return int(load().d2d_oracle_max_threads())


def full_step(pos, link_tx, link_rx, link_type, raw_actions, cols: orc.DeviceColumns, spec: orc.PathLossSpec, *,
              pwr_levels=None, min_capacity_mbps: float = 0.0, with_obs: bool = True, threads: int = 1, out=None):
    """d2d_env.py:62-71 end to end on arrays.  ``out``: a dict from a previous call whose arrays are reused (timing loops)."""
    if spec.kind != 'log_distance':
        raise ValueError('the C oracle restates the log-distance model only')
    lib = load()
    pos = np.ascontiguousarray(pos, dtype=np.float64)
    b, d, _ = pos.shape
    tx = np.ascontiguousarray(link_tx, dtype=np.int32); rx = np.ascontiguousarray(link_rx, dtype=np.int32)
    ty = np.ascontiguousarray(link_type, dtype=np.int32)
    n = len(tx)
    if pwr_levels is None:
        pwr_levels = orc.pwr_levels_for(ty)
    lv = np.ascontiguousarray(pwr_levels, dtype=np.int32)
    act = np.ascontiguousarray(raw_actions, dtype=np.int64)
    assert act.shape == (b, n)
    if out is None:
        out = {'rb': np.empty((b, n), np.int64), 'pwr': np.empty((b, n), np.int64), 'reward': np.empty(b),
               'table': np.empty((b, n, 6))}
        for k in ('sinr_db', 'snr_db', 'rate_bps', 'capacity_mbps'):
            out[k] = np.empty((b, n))
        if with_obs:
            out['obs'] = np.empty((b, n, 6 * n))
    col = [np.ascontiguousarray(c, dtype=np.float64) for c in (cols.eirp_off_db, cols.rx_off_db, cols.noise_dbm, cols.sens_dbm, cols.bw_hz)]
    a = _Args()
    a.B, a.D, a.N, a.R = b, d, n, 0
    a.pos = pos.ctypes.data_as(_DP)
    a.link_tx, a.link_rx, a.link_type, a.pwr_levels = (x.ctypes.data_as(_IP) for x in (tx, rx, ty, lv))
    a.actions = act.ctypes.data_as(_LP)
    a.eirp_off_db, a.rx_off_db, a.noise_dbm, a.sens_dbm, a.bw_hz = (c.ctypes.data_as(_DP) for c in col)
    a.ple = float(spec.ple)
    a.pl_const_db = float(orc.pl_constant_db(spec.carrier_freq_ghz, spec.ple))
    a.min_capacity_mbps = float(min_capacity_mbps)
    a.rb, a.pwr = out['rb'].ctypes.data_as(_LP), out['pwr'].ctypes.data_as(_LP)
    for k in ('sinr_db', 'snr_db', 'rate_bps', 'capacity_mbps', 'reward', 'table'):
        setattr(a, k, out[k].ctypes.data_as(_DP))
    a.obs = out['obs'].ctypes.data_as(_DP) if with_obs and 'obs' in out else None
    a.threads = int(threads)
    if lib.d2d_oracle_step(C.byref(a)) != 0:
        raise MemoryError('d2d_oracle_step failed')
    return out
