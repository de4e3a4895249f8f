#!/usr/bin/env python3
"""Fuzz of the device-side reset sampler (csrc/d2d_reset.hip) against the oracle's sampler fed the same Philox uniforms: random
seeds, episodes, env offsets, batch sizes, device counts and radii.  A position may differ by more than 2e-6 of the cell radius
only where a rejection decision sits on the cell boundary (the draw that one side kept and the other redrew lands within 1e-3 m of it).

    python tools/fuzz_reset.py [seconds]
"""
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / 'tests')]
from gym_d2d_amd.simulator import Simulator
from oracle import d2d_oracle as orc


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    rng = np.random.default_rng(int(time.time()))
    t0, cases, worst, boundary = time.time(), 0, 0.0, 0
    while time.time() - t0 < budget:
        b, cues, dues = int(rng.integers(1, 300)), int(rng.integers(0, 40)), int(rng.integers(1, 40))
        cell, d2d = float(rng.choice([50.0, 500.0, 2000.0])), float(rng.choice([5.0, 20.0, 60.0]))
        seed, episode, first = int(rng.integers(0, 2 ** 62)), int(rng.integers(0, 1000)), int(rng.integers(0, 10 ** 6))
        sim = Simulator(dict(num_cues=cues, num_due_pairs=dues, num_envs=b, cell_radius_m=cell, d2d_radius_m=d2d))
        sim.handle.set_env_offset(first)
        sim.reset_device(seed=seed, episode=episode)
        got = sim.positions().astype(np.float64)
        d = 1 + cues + 2 * dues
        u = orc.reset_uniforms(seed, episode, b, d, 64, first_env=first)
        ref, used = orc.sample_positions_from_uniforms(u, cues, dues, cell, d2d)
        dev = np.abs(got - ref).max(axis=2)
        close = dev <= cell * 2e-6
        # a rejection decision on the cell boundary: the oracle (fp64) kept a draw the kernel (fp32) redrew, or the other way round
        r_ref, r_got = np.hypot(ref[..., 0], ref[..., 1]), np.hypot(got[..., 0], got[..., 1])
        tol = 1e-3 * max(1.0, cell / 500.0)
        ok = close | (np.abs(r_ref - cell) < tol) | (np.abs(r_got - cell) < tol)
        if not ok.all():
            k = np.argwhere(~ok)[0]
            print('MISMATCH', dict(b=b, cues=cues, dues=dues, cell=cell, d2d=d2d, seed=seed, episode=episode, first=first), 'env/dev', k.tolist(),
                  got[tuple(k)], ref[tuple(k)], flush=True)
            sys.exit(1)
        worst = max(worst, float((dev[close] / cell).max()))
        boundary += int((~close).sum())
        assert (np.hypot(got[..., 0], got[..., 1]) <= cell * (1 + 1e-6)).all()
        tx = got[:, 1 + cues::2]; rx = got[:, 2 + cues::2]
        # (the receiver is drawn within d2d of its transmitter and both are then float32 absolute coordinates: half an ulp of the
        # cell radius each - 1.2e-4 m at 2000 m)
        assert (np.hypot(tx[..., 0] - rx[..., 0], tx[..., 1] - rx[..., 1]) <= d2d * (1 + 1e-5) + cell * 2.4e-7).all()
        sim.handle.close()
        cases += 1
    print(f'reset fuzz ok: {cases} random configurations, worst deviation {worst:.2e} of the cell radius, {boundary} boundary decisions', flush=True)


if __name__ == '__main__':
    main()
