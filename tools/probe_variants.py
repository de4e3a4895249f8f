#!/usr/bin/env python3
"""Print the write-ceiling probe's per-variant rates (d2d_probe_write_variants) as JSON lines."""
import json
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from gym_d2d_amd import _native

h = _native.Handle(num_envs=8, num_rbs=4, num_cues=4, num_due_pairs=4, pwr_levels_due=21, pwr_levels_cue=24, pwr_levels_mbs=47)
blocks, rows = (768, 1024, 512, 256), (2, 4, 8, 32)
for rnd in range(3):
    best, rates = h.probe_write_variants(8 << 30, 5)
    for v, r in enumerate(rates):
        name = 'hipMemsetAsync' if v == 32 else f'block {blocks[v & 3]} x {rows[(v >> 2) & 3]} rows, ' + ('plain' if v & 16 else 'nt')
        print(json.dumps({'round': rnd, 'variant': v, 'shape': name, 'GBps': round(r, 1)}))
    print(json.dumps({'round': rnd, 'best_GBps': round(best, 1)}), flush=True)
h.close()
