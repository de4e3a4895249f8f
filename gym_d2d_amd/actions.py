"""Actions of one step (the reference's gym_d2d/actions.py surface): who transmits to whom, on which resource block,
at what power.  On the GPU the same information is the link table (tx, rx, type) plus the rb / pwr arrays; these
objects exist for the dict-style single-env API and for Python plugins."""
from __future__ import annotations

from collections import UserDict
from dataclasses import dataclass
from typing import Dict, Set

from .device import Device
from .link_type import LinkType


@dataclass(frozen=True)
class Action:
    tx: Device              # transmitting device
    rx: Device              # receiving device
    link_type: LinkType     # derived from the transmitter's class (d2d_env.py:80-91)
    rb: int                 # resource block index; not range-checked, equality is what matters
    tx_pwr_dBm: float       # transmit power before antenna gains / losses


def make_action(tx: Device, rx: Device, link_type: LinkType, rb: int, tx_pwr_dBm) -> Action:
    """Action(tx, rx, link_type, rb, tx_pwr_dBm) without the five object.__setattr__ calls a frozen dataclass's
    __init__ makes (the per-step cost of the single-env API is ~N of these)."""
    act = object.__new__(Action)
    object.__setattr__(act, '__dict__', {'tx': tx, 'rx': rx, 'link_type': link_type, 'rb': rb, 'tx_pwr_dBm': tx_pwr_dBm})
    return act


class Actions(UserDict):
    """{(tx_id, rx_id): Action}; insertion order is the agent order of every output.  Keeps a per-RB index that is
    built on first use and dropped by clear()."""

    def __init__(self, *args, **kwargs) -> None:
        super().__init__(*args, **kwargs)
        self._by_rb: Dict[int, Set[Action]] = {}

    def clear(self) -> None:
        super().clear()
        self._by_rb = {}

    def get_actions_by_rb(self, rb: int) -> Set[Action]:
        """Every action that uses resource block `rb` (possibly empty)."""
        if not self._by_rb:
            for act in self.data.values():
                self._by_rb.setdefault(act.rb, set()).add(act)
        return self._by_rb.setdefault(rb, set())
