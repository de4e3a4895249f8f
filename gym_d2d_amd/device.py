"""Devices and their link budgets (mirrors gym_d2d/device.py) plus the export of the per-device SoA columns the
HIP kernels consume.

A `Device` is a config dict with property access, exactly as in the reference, so user PathLoss plugins that read
`tx.tx_antenna_gain_dBi`, `rx.position.distance(...)`, ... keep working.  What is new is `link_budget_columns()`:
the dB offsets that Device.eirp_dBm / rx_signal_level_dBm add around the transmit power and path loss are constants
of the device, so they are flattened once into float64 columns [D] and handed to d2d_set_device_table().
"""
from __future__ import annotations

from typing import Dict, Iterable, Optional

import numpy as np

from .conversion import dB_to_linear, dBm_to_W
from .id import Id
from .position import Position
from .utils import merge_dicts

# thermal noise of one 180 kHz LTE resource block (device.py:9-11)
THERMAL_NOISE_POWER_dBm = -121.45
THERMAL_NOISE_POWER_mW = dB_to_linear(THERMAL_NOISE_POWER_dBm)
THERMAL_NOISE_POWER_W = dBm_to_W(THERMAL_NOISE_POWER_dBm)

# Default link-budget tables (values: device.py:12-41).
DEFAULT_DEVICE_CONFIG = {'num_PRB': 1, 'num_subcarriers': 12, 'subcarrier_spacing_kHz': 15.0}
DEFAULT_BASE_STATION_CONFIG = {
    **DEFAULT_DEVICE_CONFIG,
    'max_tx_power_dBm': 46.0, 'antenna_height_m': 23.0,
    'tx_antenna_gain_dBi': 17.5, 'rx_antenna_gain_dBi': 17.5,
    'thermal_noise_dBm': -118.4, 'noise_figure_dB': 2.0, 'sinr_dB': -7.0,
    'ix_margin_dB': 2.0, 'cable_loss_dB': 2.0, 'masthead_amplifier_gain_dB': 2.0,
}
DEFAULT_UE_CONFIG = {
    **DEFAULT_DEVICE_CONFIG,
    'max_tx_power_dBm': 23.0, 'antenna_height_m': 1.5,
    'tx_antenna_gain_dBi': 0.0, 'rx_antenna_gain_dBi': 0.0,
    'thermal_noise_dBm': -104.5, 'noise_figure_dB': 7.0, 'sinr_dB': -10.0,
    'ix_margin_dB': 3.0, 'control_channel_overhead_dB': 1.0, 'body_loss_dB': 3.0,
}


def _cfg_property(key: str, cast=None):
    def getter(self):
        value = self.config[key]
        return cast(value) if cast else value
    getter.__name__ = key
    return property(getter)


class Device:
    """Common part of base stations and user equipment."""
    _defaults: Dict[str, float] = DEFAULT_DEVICE_CONFIG

    def __init__(self, id_, config: Optional[dict] = None) -> None:
        self.id = Id(id_)
        self.config: dict = merge_dicts(dict(self._defaults), config or {})
        self.position: Position = Position(0, 0)

    # ---- dB offsets: the only thing the device side needs from a device
    def tx_offset_dB(self) -> float:
        """eirp_dBm(p) - p   (device.py:51-60)."""
        return self.tx_antenna_gain_dBi - self.ix_margin_dB

    def rx_offset_dB(self) -> float:
        """rx_signal_level_dBm(eirp, pl) - (eirp - pl)   (device.py:62-72)."""
        return self.rx_antenna_gain_dBi

    def eirp_dBm(self, tx_pwr_dBm: float) -> float:
        """Effective isotropically radiated power for a transmit power."""
        return tx_pwr_dBm + self.tx_offset_dB()

    def rx_signal_level_dBm(self, eirp_dBm: float, path_loss_dB: float) -> float:
        """Signal level at this receiver for a transmitted EIRP and a path loss."""
        return eirp_dBm - path_loss_dB + self.rx_offset_dB()

    @property
    def rx_noise_floor_dBm(self) -> float:
        return self.noise_figure_dB + self.thermal_noise_dBm

    @property
    def rx_sensitivity_dBm(self) -> float:
        return self.rx_noise_floor_dBm + self.sinr_dB

    @property
    def rb_bandwidth_kHz(self) -> int:
        return self.num_subcarriers * self.subcarrier_spacing_kHz

    def set_position(self, pos: Position) -> None:
        self.position = pos

    num_subcarriers = _cfg_property('num_subcarriers', int)
    subcarrier_spacing_kHz = _cfg_property('subcarrier_spacing_kHz', int)
    max_tx_power_dBm = _cfg_property('max_tx_power_dBm')
    antenna_height_m = _cfg_property('antenna_height_m')
    tx_antenna_gain_dBi = _cfg_property('tx_antenna_gain_dBi')
    rx_antenna_gain_dBi = _cfg_property('rx_antenna_gain_dBi')
    noise_figure_dB = _cfg_property('noise_figure_dB')
    thermal_noise_dBm = _cfg_property('thermal_noise_dBm')
    sinr_dB = _cfg_property('sinr_dB')
    ix_margin_dB = _cfg_property('ix_margin_dB')


class BaseStation(Device):
    """Macro base station: cable loss and masthead amplifier on both directions (device.py:130-151)."""
    _defaults = DEFAULT_BASE_STATION_CONFIG
    cable_loss_dB = _cfg_property('cable_loss_dB')
    masthead_amplifier_gain_dB = _cfg_property('masthead_amplifier_gain_dB')

    def tx_offset_dB(self) -> float:
        return super().tx_offset_dB() - self.cable_loss_dB + self.masthead_amplifier_gain_dB

    def rx_offset_dB(self) -> float:
        return super().rx_offset_dB() - self.cable_loss_dB + self.masthead_amplifier_gain_dB

    def __repr__(self) -> str:
        return f'<BS:{self.id}>'


class UserEquipment(Device):
    """Handset: body loss on both directions (device.py:154-173)."""
    _defaults = DEFAULT_UE_CONFIG
    control_channel_overhead_dB = _cfg_property('control_channel_overhead_dB')
    body_loss_dB = _cfg_property('body_loss_dB')

    def tx_offset_dB(self) -> float:
        return super().tx_offset_dB() - self.body_loss_dB

    def rx_offset_dB(self) -> float:
        return super().rx_offset_dB() - self.body_loss_dB

    def __repr__(self) -> str:
        return f'<UE:{self.id}>'


def link_budget_columns(devices: Iterable[Device]) -> Dict[str, np.ndarray]:
    """Flatten devices (in index order) into the float64 columns of d2d_set_device_table()."""
    devs = list(devices)
    return {
        'eirp_off_db': np.array([d.tx_offset_dB() for d in devs], dtype=np.float64),
        'rx_off_db': np.array([d.rx_offset_dB() for d in devs], dtype=np.float64),
        'noise_dbm': np.array([d.thermal_noise_dBm for d in devs], dtype=np.float64),     # simulator.py:107,115
        'sens_dbm': np.array([d.rx_sensitivity_dBm for d in devs], dtype=np.float64),     # simulator.py:123,149
        'bw_hz': np.array([d.rb_bandwidth_kHz * 1000 for d in devs], dtype=np.float64),   # simulator.py:150
    }
