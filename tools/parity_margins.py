#!/usr/bin/env python3
"""Worst |gpu - oracle| / max(|oracle|, 1) over random scenarios, per path-loss model, position precision and output - the margins
behind the 1e-5 bar.  Random sizes, RB counts, link subsets are the property test's business
(tests/test_gpu_property.py); this run is about the ARITHMETIC: every path-loss mode the kernels have (1 / d^2, the one-integer-k
power law for every k, the general split where exponents share no k, COST-Hata with per-device antenna heights) on float32
layouts and on float64 layouts through d2d_set_positions_f64.

    python tools/parity_margins.py [seconds] [--out profiles/rN_parity_margins.json]
"""
import json
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / 'tests')]
from gym_d2d_amd import _native
from gym_d2d_amd.path_loss import AreaType, CostHataPathLoss, LogDistancePathLoss
from gym_d2d_amd.simulator import Simulator
from oracle import d2d_oracle as orc
from sim_util import default_links

FIELDS = (('sinr_db', _native.BUF_SINR_DB), ('snr_db', _native.BUF_SNR_DB), ('rate_bps', _native.BUF_RATE_BPS), ('capacity_mbps', _native.BUF_CAPACITY))


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 and not sys.argv[1].startswith('--') else 120.0
    out_path = sys.argv[sys.argv.index('--out') + 1] if '--out' in sys.argv else ''
    rng = np.random.default_rng(int(time.time()))
    worst, count = {}, {}
    t0 = time.time()
    while time.time() - t0 < budget:
        model = str(rng.choice(['log2', 'ple_k', 'ple_k', 'hata_urban', 'hata_suburban', 'hata_mixed_heights']))
        cues, dues = int(rng.choice([0, 6, 25, 64])), int(rng.choice([7, 25, 64]))
        rbs = int(rng.choice([2, 8, 25, 64]))
        b = int(rng.integers(2, 24))
        cfg = dict(num_rbs=rbs, num_cues=cues, num_due_pairs=dues, num_envs=b)
        ids, cfgs, is_bs = orc.device_configs(cues, dues)
        tag = model
        if model == 'ple_k':
            ple = float(rng.uniform(0.6, 8.4))
            tag = f'ple (k = {int(round(ple))})'

            class Ple(LogDistancePathLoss):
                def __init__(self, f):
                    super().__init__(f, ple=ple)
            cfg['path_loss_model'] = Ple
            spec = orc.PathLossSpec('log_distance', 2.1, ple=ple)
        elif model.startswith('hata'):
            area = 'urban' if model == 'hata_urban' else 'suburban'

            class Hata(CostHataPathLoss):
                def __init__(self, f):
                    super().__init__(f, AreaType.URBAN if area == 'urban' else AreaType.SUBURBAN)
            cfg['path_loss_model'] = Hata
            spec = orc.PathLossSpec('cost_hata', 2.1, area=area)
            if model == 'hata_mixed_heights':
                # antenna heights per device: transmitters' slopes then straddle integers (no common k: the general split) or not
                import tempfile
                over = {}
                for k in rng.choice(len(ids), size=min(len(ids), 6), replace=False):
                    h = float(np.round(rng.uniform(20.0, 90.0) if is_bs[k] else rng.uniform(1.0, 9.0), 2))   # (a receiving antenna above ~100 m leaves the float32 linear range: refused by the library)
                    base = {'num_subcarriers': 12, 'subcarrier_spacing_kHz': 15, 'antenna_height_m': h}
                    if not is_bs[k]:
                        base['max_tx_power_dBm'] = 23 if ids[k].startswith('cue') else 20
                    over[ids[k]] = {'position': [0.0, 0.0], 'config': base}
                path = Path(tempfile.mkdtemp()) / 'devices.json'
                path.write_text(json.dumps(over))
                cfg['device_config_file'] = path
                _, cfgs, _ = orc.device_configs(cues, dues, overrides=over)
        else:
            spec = orc.PathLossSpec()
        cols = orc.device_columns(cfgs, is_bs)
        sim = Simulator(cfg, max_links=cues + dues)
        sim.set_links(sim.default_link_keys())
        d = 1 + cues + 2 * dues
        pos, _ = orc.sample_positions_from_uniforms(rng.random((b, d, 32, 2)), cues, dues, 500.0, 20.0)
        precision = 'float64 positions' if rng.random() < 0.5 else 'float32 positions'
        if precision.startswith('float32'):
            pos = pos.astype(np.float32).astype(np.float64)
        sim.set_positions(pos)                     # float64 array: hi + lo pairs where float32 cannot hold the values
        p = sim.config.num_pwr_actions
        raw = np.concatenate([rng.integers(0, rbs * p['cue'], (b, cues)), rng.integers(0, rbs * p['due'], (b, dues))], 1).astype(np.int32)
        h = sim.handle
        h.set_obs_mode(int(rng.choice([_native.OBS_TABLE, _native.OBS_NONE])))
        sim.step_arrays(raw)
        if sim.handle.status_flags() & _native.FLAG_ZERO_DISTANCE:
            h.close()
            continue
        tx, rx, ty = default_links(cues, dues)
        ref = orc.full_step(pos, tx, rx, ty, raw, cols, spec, with_obs=False)
        for f, buf in FIELDS:
            got = sim.fetch(buf).astype(np.float64)
            e = float(np.max(np.abs(got - ref[f]) / np.maximum(np.abs(ref[f]), 1.0)))
            if not np.isfinite(e):
                raise SystemExit(f'non-finite result: {tag} / {precision} / {f} ' + json.dumps({k: str(v) for k, v in cfg.items()}))
            key = f'{tag} / {precision} / {f}'
            worst[key] = max(worst.get(key, 0.0), e)
        count[f'{tag} / {precision}'] = count.get(f'{tag} / {precision}', 0) + 1
        h.close()
    out = {'what': 'worst |gpu - oracle| / max(|oracle|, 1) over random batches (tools/parity_margins.py), per path-loss model, position '
                   'precision and output; bar 1e-5', 'seconds': budget, 'scenarios': dict(sorted(count.items())),
           'worst': {k: float(f'{v:.3e}') for k, v in sorted(worst.items())}, 'worst_overall': max(worst.values())}
    print(json.dumps(out, indent=1))
    if out_path:
        Path(out_path).write_text(json.dumps(out, indent=1))
    assert out['worst_overall'] <= 1e-5, out['worst_overall']


if __name__ == '__main__':
    main()
