#!/usr/bin/env python3
"""Pure fills of 64 MiB (about the 61 MB of obs BASELINE config 2 writes per step), for a rocprofv3 kernel trace: what does a
write burst of that size cost as a KERNEL (d2d_probe_write_variants' own figure is group timed and includes the launch gaps)?"""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
from gym_d2d_amd import _native

h = _native.Handle(num_envs=8, num_rbs=4, num_cues=4, num_due_pairs=4, pwr_levels_due=21, pwr_levels_cue=24, pwr_levels_mbs=47)
for size in (64 << 20, 128 << 20):
    best, rates = h.probe_write_variants(size, 20)
    print(size >> 20, 'MiB: best', round(best), 'GB/s group timed; obs geometry', round(rates[0]), '; memset', round(rates[32]))
h.close()
