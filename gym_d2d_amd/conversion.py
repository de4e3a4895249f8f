"""Host-side decibel helpers with the reference's names (gym_d2d/conversion.py:4-33).

The simulation path does NOT go through these (it runs in the linear domain on the GPU); they are kept because
user plugins written against the reference import them.
"""
import math

_TEN = 10.0


def dB_to_linear(dB: float) -> float:
    """Ratio in decibels -> plain ratio."""
    return math.pow(_TEN, dB / _TEN)


def linear_to_dB(linear: float) -> float:
    """Plain ratio -> decibels (raises ValueError for 0, like math.log10)."""
    return _TEN * math.log10(linear)


def dBm_to_W(dBm: float) -> float:
    return dB_to_linear(dBm) * 1e-3


def W_to_dBm(watts: float) -> float:
    return linear_to_dB(watts * 1e3)
