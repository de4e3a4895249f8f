"""Link direction of an action (mirrors gym_d2d/link_type.py:4-7).  The integer values are the ones the HIP
kernels branch on (include/d2d_hip.h d2d_link_type)."""
from enum import Enum


class LinkType(Enum):
    UPLINK = 1      # CUE -> base station
    DOWNLINK = 2    # base station -> CUE
    SIDELINK = 3    # DUE tx -> DUE rx
