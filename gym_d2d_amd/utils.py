"""Small helpers shared by the host-side mirror."""
from __future__ import annotations


def merge_dicts(original: dict, other: dict) -> dict:
    """Recursively overlay `other` onto `original` IN PLACE and return it (gym_d2d/utils.py:1-16 semantics:
    nested dicts are merged key by key, everything else is overwritten)."""
    for key, value in other.items():
        if isinstance(value, dict) and isinstance(original.get(key), dict):
            merge_dicts(original[key], value)
        else:
            original[key] = value
    return original
