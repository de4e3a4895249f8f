#!/usr/bin/env python3
"""Condense the rocprofv3 output of tools/profile_bench.sh into the small summaries kept in profiles/:

    python tools/summarize_profiles.py <tag> <workload> <obs> <raw dir>

  profiles/<tag>_kernel_stats_<workload>_<obs>.csv   kernel-trace --stats table (names shortened)
  profiles/<tag>_pmc_<workload>_<obs>.json           per-kernel HBM bytes per launch (WRITE_SIZE + 2 * FETCH_SIZE, separate
                                                     passes) and the SQ counters per wave, stamped with the digest of the
                                                     kernel sources they were collected on (bench.py only quotes `traffic`
                                                     from a summary whose digest matches the running library)
"""
import collections
import csv
import glob
import json
import os
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


def counters(raw, sub):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(f'{raw}/{sub}/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            if 'd2d::' in r['Kernel_Name']:
                acc[r['Kernel_Name'][:70]][r['Counter_Name']].append(float(r['Counter_Value']))
    return acc


def durations(raw, skip_first):
    """Per d2d kernel, from the kernel trace of the --stats run: launch durations in dispatch order with the first
    `skip_first` launches of every kernel (warm-up, page touching) left out - median and mean of the rest beside rocprofv3's
    own all-launch average."""
    per = collections.defaultdict(list)
    for f in glob.glob(f'{raw}/kt/**/*kernel_trace.csv', recursive=True):
        rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
        for r in rows:
            if 'd2d::' in r['Kernel_Name']:
                per[r['Kernel_Name'][:70]].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
    out = {}
    for k, v in per.items():
        rest = sorted(v[skip_first:]) if len(v) > 2 * skip_first else sorted(v)
        out[k] = {'launches': len(v), 'warmup_launches_excluded': len(v) - len(rest), 'median_us': round(rest[len(rest) // 2], 3),
                  'mean_us': round(sum(rest) / len(rest), 3), 'min_us': round(rest[0], 3), 'max_us': round(rest[-1], 3),
                  'mean_us_all_launches': round(sum(v) / len(v), 3)}
    return out


def main():
    tag, wl, obs, raw = sys.argv[1:5]
    extra = ' '.join(sys.argv[5:])                    # extra bench arguments the passes ran with, e.g. --no-export
    # one summary per CONFIGURATION: the compact modes with / without the decoded (rb, pwr) planes and with the per-env reward
    # differ in their bytes per link (file names of rounds 1-3: the table mode's summary is the --no-export one)
    suffix = ''
    key_obs = obs
    if 'float64' in extra:
        suffix += '_f64'; key_obs += '_f64'
    if '--reward-per-env' in extra:
        suffix += '_per_env_reward'; key_obs += '_per_env_reward'
    if obs != 'linear' and '--no-export' not in extra:
        suffix += '_export'
    from gym_d2d_amd.build import source_digest
    stats = glob.glob(f'{raw}/kt/**/*kernel_stats.csv', recursive=True)
    if stats:
        rows = list(csv.reader(open(stats[0])))
        out = ROOT / 'profiles' / f'{tag}_kernel_stats_{wl}_{obs}{suffix}.csv'
        with open(out, 'w', newline='') as f:
            w = csv.writer(f)
            for r in rows:
                r[0] = r[0][:90]                       # torch's templated kernel names run to kilobytes
                w.writerow(r)
        print(out.read_text()[:700])
    kernels = {}
    for sub, name in (('pmc_write', 'WRITE_SIZE'), ('pmc_fetch', 'FETCH_SIZE')):
        for k, d in counters(raw, sub).items():
            v = d.get(name, [])
            if v:
                kernels.setdefault(k, {})[name + '_KiB_mean_per_launch'] = sum(v) / len(v)
                kernels[k][name + '_launches'] = len(v)
    for d in kernels.values():
        w_b = d.get('WRITE_SIZE_KiB_mean_per_launch', 0.0) * 1024
        f_b = d.get('FETCH_SIZE_KiB_mean_per_launch', 0.0) * 1024
        # gfx950: FETCH_SIZE reports half the bytes of a coalesced stream (MI355X_MICROARCH.md, HBM) -> doubled
        d['hbm_bytes_per_launch'] = w_b + 2 * f_b
    for sub in ('pmc_sq1', 'pmc_sq2'):
        for k, d in counters(raw, sub).items():
            sq = kernels.setdefault(k, {}).setdefault('sq_counters_mean_per_launch', {})
            for c, v in d.items():
                sq[c] = sum(v) / len(v)
    for d in kernels.values():
        sq = d.get('sq_counters_mean_per_launch')
        if sq and sq.get('SQ_WAVES'):
            d['sq_per_wave'] = {c: round(v / sq['SQ_WAVES'], 2) for c, v in sq.items() if c != 'SQ_WAVES'}
            if sq.get('SQ_LDS_IDX_ACTIVE'):
                d['lds_bank_conflict_ratio'] = round(sq.get('SQ_LDS_BANK_CONFLICT', 0.0) / sq['SQ_LDS_IDX_ACTIVE'], 3)
            if sq.get('SQ_WAVE_CYCLES'):
                d['wait_any_fraction_of_wave_cycles'] = round(sq.get('SQ_WAIT_ANY', 0.0) / sq['SQ_WAVE_CYCLES'], 3)
    rec = {'command': 'tools/profile_bench.sh: rocprofv3 --pmc <one counter set per pass> --output-format csv -- python3 bench.py '
                      f'--workload {wl} --obs {obs} --no-cpu-baseline --no-single-env-latency --no-extras {extra} --steps 5 --warmup 1'.replace('  ', ' '),
           'workload_key': f'{wl}/{key_obs}', 'source_digest': source_digest(), 'kernels': kernels,
           'kernel_trace_durations': durations(raw, int(os.environ.get('PROFILE_WARMUP', '10')) + 2)}
    out = ROOT / 'profiles' / f'{tag}_pmc_{wl}_{obs}{suffix}.json'
    out.write_text(json.dumps(rec, indent=1))
    if (wl, obs) == ('stress', 'table') and suffix == '':
        # the step kernel is 100 % of this mode's step: its own summary under the name VERDICT r1 asked for
        step = {k: d for k, d in kernels.items() if 'step_kernel' in k or 'rollout_kernel' in k}
        for k, d in step.items():
            sq = d.get('sq_per_wave', {})
            d['summary'] = {'SQ_WAIT_ANY / SQ_WAVE_CYCLES': d.get('wait_any_fraction_of_wave_cycles'),
                            'SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE': d.get('lds_bank_conflict_ratio'),
                            'VALU per wave': sq.get('SQ_INSTS_VALU'), 'SALU per wave': sq.get('SQ_INSTS_SALU'),
                            'LDS per wave': sq.get('SQ_INSTS_LDS'), 'VMEM rd / wr per wave': [sq.get('SQ_INSTS_VMEM_RD'), sq.get('SQ_INSTS_VMEM_WR')],
                            'FETCH (x2 corrected) / WRITE bytes per launch': [2 * 1024 * d.get('FETCH_SIZE_KiB_mean_per_launch', 0.0),
                                                                             1024 * d.get('WRITE_SIZE_KiB_mean_per_launch', 0.0)],
                            'algorithmic bytes per launch (4096 x 512 x 64)': 4096 * 512 * 64}
        (ROOT / 'profiles' / f'{tag}_pmc_step_kernel.json').write_text(json.dumps(dict(rec, kernels=step), indent=1))
    for k, d in rec['kernel_trace_durations'].items():
        print(k[:60], d)
    for k, d in kernels.items():
        print(k[:60], {a: (round(b) if isinstance(b, float) else b) for a, b in d.items() if not isinstance(b, dict)})


if __name__ == '__main__':
    main()
