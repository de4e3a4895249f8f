"""libd2d_probe.so (include/d2d_hip_diag.h): the write-ceiling probe - measurement equipment, not the product library."""
from pathlib import Path

import numpy as np
import pytest

from golden_util import load_case, rel_err
from oracle import d2d_oracle as orc
from sim_util import OUTS, assert_same as _same, default_links, random_batch as _batch, random_layout, search_variants as _variants, snapshot as _snapshot

ROOT = Path(__file__).resolve().parent.parent
pytestmark = pytest.mark.gpu
TOL = 1e-5


def test_write_ceiling_probe_family(native):
    """libd2d_probe.so (include/d2d_hip_diag.h; measurement equipment, not the product library): every variant of the fill
    family reports a plausible rate, the best is the maximum, the obs kernel's own geometry is variant 0."""
    import sys
    sys.path.insert(0, str(ROOT / 'tools'))
    import write_probe
    best, rates = write_probe.write_variants(1 << 30, 3)
    assert len(rates) == 33 and all(500.0 < r < 8000.0 for r in rates), rates
    assert abs(best - max(rates)) < 1e-6
    with pytest.raises(ValueError):
        write_probe.write_variants(1 << 20, 1)              # below one group of regions


def test_staged_write_probe(native):
    """libd2d_probe.so's staged forms: the fill family with an LDS stage + barrier and / or a per-wave sleep stagger in front of
    the stores - runs, writes what it says, refuses nonsense."""
    import sys
    sys.path.insert(0, str(ROOT / 'tools'))
    import write_probe
    for variant, stagger in ((0, 0), (32, 0), (64, 2), (96, 1), (32 + 1, 0), (128, 0), (256 + 32, 0), (384, 0), (512 + 1, 0)):
        assert write_probe.write_staged(64 << 20, variant, stagger, iters=2) > 100.0
    import torch
    dst = torch.zeros(64 << 18, dtype=torch.float32, device='cuda')       # 64 MiB
    assert write_probe.write_staged(64 << 20, 32, 0, iters=1, dst_ptr=dst.data_ptr()) > 100.0
    with pytest.raises(ValueError):
        write_probe.write_staged(1 << 20, 0)
    with pytest.raises(ValueError):
        write_probe.write_staged(64 << 20, 640)
