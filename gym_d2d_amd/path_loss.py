"""PathLoss plugins (mirrors gym_d2d/path_loss.py) and their lowering to the HIP kernels.

Plugin contract (unchanged from the reference): a PathLoss class is passed in env_config['path_loss_model'],
is constructed with one positional argument (carrier_freq_GHz) and is callable as model(tx, rx) -> dB.

Lowering: every built-in model is a power law in distance,
    PL_dB(tx, rx) = a_tx[tx] + a_rx[rx] + 10 * n[tx] * log10(d_m),
so `power_law_columns(devices)` returns the three per-device float64 columns for d2d_set_path_loss_power_law()
and the kernel evaluates gain = 10^(-(a_tx+a_rx)/10) * d^-n in the linear domain (one v_rcp per pair when n == 2).
A user subclass that only defines __call__ is evaluated on the host once per episode into a [D,D] table
(`table_db`, d2d_set_path_loss_table) - legal because positions are static between resets (simulator.py:61-75).
For batches a per-object __call__ is B x N x N Python calls per reset (1.1e9 at 4096 x 512): `ArrayPathLoss` is the
array-native form of the same plugin - compute(view) -> pl_db[B,N,N] on whole arrays (torch CUDA tensors in a batch, NumPy
for a single pair), handed to the library from device memory (d2d_set_path_loss_link_table_dev).
"""
from __future__ import annotations

import math
from abc import ABC, abstractmethod
from enum import Enum
from random import gauss
from typing import Dict, Optional, Sequence

import numpy as np

from .device import Device

SPEED_OF_LIGHT = 299792458  # m/s


class PathLoss(ABC):
    def __init__(self, carrier_freq_GHz: float) -> None:
        self.carrier_freq_GHz = float(carrier_freq_GHz)

    @abstractmethod
    def __call__(self, tx: Device, rx: Device) -> float:
        """Path loss in dB from `tx` to `rx`."""

    # ---- lowering hooks -------------------------------------------------------------------------------
    def power_law_columns(self, devices: Sequence[Device]) -> Optional[Dict[str, np.ndarray]]:
        """Per-device columns {'a_tx_db', 'a_rx_db', 'exponent'} if the model is a power law in distance,
        else None (the table route is used)."""
        return None

    def table_db(self, devices: Sequence[Device], tx_indices: Optional[Sequence[int]] = None,
                 rx_indices: Optional[Sequence[int]] = None) -> np.ndarray:
        """Host evaluation of the model: out[tx_index, rx_index] in dB (float64, as the plugin returns it).  Only the pairs
        (tx in tx_indices) x (rx in rx_indices) are evaluated - the transmitters and receivers of the links that act: the
        step reads PL(tx of link j, rx of link i) and nothing else (simulator.py:93,100,114), which at 256 CUE + 256 DUE
        pairs is 512 x 257 calls instead of 769 x 769.  None = every device.  Everything else stays NaN, as does a pair the
        model cannot evaluate (zero distance -> ValueError in math.log10); the env raises if such a pair is ever used,
        which is when the reference would have raised."""
        devs = list(devices)
        out = np.full((len(devs), len(devs)), np.nan, dtype=np.float64)
        txs = range(len(devs)) if tx_indices is None else sorted(set(int(i) for i in tx_indices))
        rxs = range(len(devs)) if rx_indices is None else sorted(set(int(j) for j in rx_indices))
        for i in txs:
            tx = devs[i]
            for j in rxs:
                if i == j:
                    continue
                try:
                    out[i, j] = self(tx, devs[j])
                except (ValueError, ZeroDivisionError):
                    pass
        return out


class PathLossView:
    """What ArrayPathLoss.compute sees: the transmitter and receiver coordinates of the links that act, as arrays, plus the
    Device objects for their static attributes.

        xp                    the array module: torch (batch on the GPU) or numpy (a single pair, or no torch)
        tx_x, tx_y            [B, N]  transmitter of link j            rx_x, rx_y   [B, N]  receiver of link i
        tx_devices, rx_devices   length-N lists of Device (antenna gains, heights, ... - the attributes device.py:85-173 exposes)
        distance()            [B, N, N] float64   d[b, j, i] = |tx of link j - rx of link i| (position.py:11-12)
        tx_column(fn) / rx_column(fn)   fn(device) per transmitter / receiver as an array shaped [1, N, 1] / [1, 1, N]

    pl_db[b, j, i] = PathLoss(tx of link j, rx of link i): j == i is the link's own signal path (simulator.py:93,114), j != i
    an interferer's (simulator.py:97-101) - the pairs d2d_set_path_loss_link_table names."""

    def __init__(self, xp, tx_x, tx_y, rx_x, rx_y, tx_devices, rx_devices, like=None):
        self.xp = xp
        self.tx_x, self.tx_y, self.rx_x, self.rx_y = tx_x, tx_y, rx_x, rx_y
        self.tx_devices, self.rx_devices = list(tx_devices), list(rx_devices)
        self._like = like if like is not None else tx_x

    def _f64(self, a):
        return a.double() if self.xp.__name__ == 'torch' else np.asarray(a, dtype=np.float64)

    def distance(self):
        dx = self._f64(self.tx_x)[:, :, None] - self._f64(self.rx_x)[:, None, :]       # float32 coordinates subtract exactly in float64
        dy = self._f64(self.tx_y)[:, :, None] - self._f64(self.rx_y)[:, None, :]
        return self.xp.sqrt(dx * dx + dy * dy)

    def _column(self, devices, fn, shape):
        vals = np.array([float(fn(d)) for d in devices], dtype=np.float64).reshape(shape)
        if self.xp.__name__ == 'torch':
            return self.xp.as_tensor(vals, device=self._like.device)
        return vals

    def tx_column(self, fn):
        return self._column(self.tx_devices, fn, (1, -1, 1))

    def rx_column(self, fn):
        return self._column(self.rx_devices, fn, (1, 1, -1))


class ArrayPathLoss(PathLoss):
    """Array-native PathLoss plugin (the batched counterpart of path_loss.py:12-25, as ArrayObsFunction is of ObsFunction):
    subclasses define compute(view) -> pl_db [B, N, N] with the array module `view.xp`.  The per-object call the reference's
    contract asks for (`model(tx, rx) -> dB`) is derived from it - one pair through the same compute - so the single-env D2DEnv
    and a batch run the same formula:

        class TwoSlope(ArrayPathLoss):
            def compute(self, view):
                d = view.distance()
                near = 20 * view.xp.log10(d) + 38.0
                far = 35 * view.xp.log10(d) - 15 * math.log10(50.0) + 38.0
                return view.xp.where(d <= 50.0, near, far) - view.tx_column(lambda t: t.tx_antenna_gain_dBi)

    A batch evaluates it once per reset on the GPU (torch) and hands the result to the library from device memory
    (d2d_set_path_loss_link_table_dev): no host table, no Python loop."""

    @abstractmethod
    def compute(self, view: PathLossView):
        """pl_db[b, j, i] in dB for the transmitter of link j and the receiver of link i, shape [B, N, N]."""

    def __call__(self, tx: Device, rx: Device) -> float:
        one = lambda v: np.array([[float(v)]], dtype=np.float64)
        view = PathLossView(np, one(tx.position.x), one(tx.position.y), one(rx.position.x), one(rx.position.y), [tx], [rx])
        out = np.asarray(self.compute(view), dtype=np.float64).reshape(-1)
        if out.size != 1 or not np.isfinite(out[0]):
            raise ValueError('math domain error')          # what math.log10(0) raises in a per-object model (path_loss.py:66)
        return float(out[0])


def pl_constant_dB(carrier_freq_GHz: float, ple: float) -> float:
    """Distance-independent part of the log-distance model: 10 n log10(f_Hz) + 10 n log10(4 pi / c)."""
    scale = 10 * ple
    return scale * math.log10(carrier_freq_GHz * 1e9) + scale * math.log10((4 * math.pi) / SPEED_OF_LIGHT)


class LogDistancePathLoss(PathLoss):
    """PL = 10 n log10(d) + 10 n log10(f) + 10 n log10(4 pi / c)   (path_loss.py:42-66)."""

    def __init__(self, carrier_freq_GHz: float, ple=2.0) -> None:
        super().__init__(carrier_freq_GHz)
        self.ple = float(ple)   # 2.0 = free space, ~3.5 = cluttered
        self.pl_constant_dB = pl_constant_dB(carrier_freq_GHz, ple)

    def _log_distance_path_loss(self, dist_m: float) -> float:
        return 10 * self.ple * math.log10(dist_m) + self.pl_constant_dB

    def __call__(self, tx: Device, rx: Device) -> float:
        return self._log_distance_path_loss(tx.position.distance(rx.position))

    def power_law_columns(self, devices):
        if type(self).__call__ not in (LogDistancePathLoss.__call__, getattr(ShadowingPathLoss, '__call__', None)):
            return None     # a subclass changed the formula: evaluate it on the host instead
        n = len(devices)
        return {'a_tx_db': np.full(n, self.pl_constant_dB), 'a_rx_db': np.zeros(n), 'exponent': np.full(n, self.ple)}


class FreeSpacePathLoss(LogDistancePathLoss):
    """Free-space (Friis) loss = log-distance with exponent 2 (the reference's own comment, path_loss.py:45).
    Not present in the reference snapshot; named by BASELINE.json config 4."""

    def __init__(self, carrier_freq_GHz: float) -> None:
        super().__init__(carrier_freq_GHz, ple=2.0)


class ShadowingPathLoss(LogDistancePathLoss):
    """Log-distance with log-normal shadowing beyond a close-in distance (path_loss.py:69-81).  A fresh Gaussian
    is drawn on every call, so this model cannot be frozen into a table."""

    def __init__(self, carrier_freq_GHz: float, ple=2.0, d0_m=100.0, chi_dB=2.7) -> None:
        super().__init__(carrier_freq_GHz, ple)
        self.d0_m = float(d0_m)
        self.chi_dB = float(chi_dB)

    def __call__(self, tx: Device, rx: Device) -> float:
        d = tx.position.distance(rx.position)
        if d <= self.d0_m:
            return self._log_distance_path_loss(d)
        anchor = self._log_distance_path_loss(self.d0_m)
        return anchor + 10 * self.ple * math.log10(d / self.d0_m) + gauss(0, self.chi_dB)

    def power_law_columns(self, devices):
        if type(self).__call__ is not ShadowingPathLoss.__call__:
            return None
        cols = LogDistancePathLoss.power_law_columns(self, devices) if \
            type(self)._log_distance_path_loss is LogDistancePathLoss._log_distance_path_loss else None
        if cols is not None:
            # LD(d0) + 10 n log10(d/d0) == LD(d): the deterministic part is the plain log-distance law; the kernel
            # adds the per-evaluation Gaussian beyond d0 (csrc/d2d_step.hip, PL_SHADOW)
            cols['shadowing'] = {'d0_m': self.d0_m, 'chi_dB': self.chi_dB}
        return cols


class AreaType(Enum):
    RURAL = 0
    SUBURBAN = 1
    URBAN = 2


class CostHataPathLoss(PathLoss):
    """COST-231 Hata (path_loss.py:90-123):
       L = 46.3 + 33.9 log10(f_MHz) - 13.82 log10(h_tx) - a(h_rx) + (44.9 - 6.55 log10(h_tx)) log10(d_km) + C."""

    def __init__(self, carrier_freq_GHz: float, area_type=AreaType.SUBURBAN) -> None:
        super().__init__(carrier_freq_GHz)
        self.area_type: AreaType = area_type

    def _ms_h_correction(self, f: float, h_rx: float) -> float:
        """Mobile-station antenna height correction a(h_rx) for carrier f in MHz."""
        if self.area_type != AreaType.URBAN:
            return (1.1 * math.log10(f) - 0.7) * h_rx - (1.56 * math.log10(f) - 0.8)
        if f >= 200:
            return 8.29 * math.log10(1.54 * h_rx) ** 2 - 1.1
        return 3.2 * math.log10(11.75 * h_rx) ** 2 - 4.97

    def _slope(self, h_tx: float) -> float:
        return 44.9 - 6.55 * math.log10(h_tx)

    def __call__(self, tx: Device, rx: Device) -> float:
        f = self.carrier_freq_GHz * 1000
        d_km = tx.position.distance(rx.position) / 1000
        h_tx, h_rx = tx.antenna_height_m, rx.antenna_height_m
        metro = 3 if self.area_type == AreaType.URBAN else 0
        return (46.3 + 33.9 * math.log10(f) - 13.82 * math.log10(h_tx) - self._ms_h_correction(f, h_rx)
                + self._slope(h_tx) * math.log10(d_km) + metro)

    def power_law_columns(self, devices):
        if type(self).__call__ is not CostHataPathLoss.__call__:
            return None
        f = self.carrier_freq_GHz * 1000
        metro = 3 if self.area_type == AreaType.URBAN else 0
        a_tx, a_rx, expo = [], [], []
        for dev in devices:
            h = dev.antenna_height_m
            slope = self._slope(h)
            # log10(d_km) = log10(d_m) - 3: fold the -3*slope into the tx constant
            a_tx.append(46.3 + 33.9 * math.log10(f) - 13.82 * math.log10(h) + metro - 3.0 * slope)
            a_rx.append(-self._ms_h_correction(f, h))
            expo.append(slope / 10.0)
        return {'a_tx_db': np.array(a_tx), 'a_rx_db': np.array(a_rx), 'exponent': np.array(expo)}
