#!/usr/bin/env python3
"""Condense rocprofv3 output under gpurun_out/ into the small summaries kept in profiles/.

    python tools/summarize_profiles.py <tag> <kernel_stats.csv> <pmc_write counter_collection.csv> <pmc_fetch ...csv>
"""
import collections
import csv
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent


def main():
    tag, stats, pmc_w, pmc_f = sys.argv[1:5]
    rows = list(csv.reader(open(stats)))
    out = ROOT / 'profiles' / f'{tag}_kernel_stats_bench_stress_steps100.csv'
    with open(out, 'w', newline='') as f:
        w = csv.writer(f)
        for r in rows:
            r[0] = r[0][:90]                       # torch's templated kernel names run to kilobytes
            w.writerow(r)
    kernels = {}
    for name, path in (('WRITE_SIZE', pmc_w), ('FETCH_SIZE', pmc_f)):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(path)):
            if r['Counter_Name'] == name and 'd2d::' in r['Kernel_Name']:
                acc[r['Kernel_Name']].append(float(r['Counter_Value']))
        for k, v in acc.items():
            kernels.setdefault(k, {})[name + '_KiB_mean_per_launch'] = sum(v) / len(v)
            kernels[k][name + '_launches'] = len(v)
    for d in kernels.values():
        w_b = d.get('WRITE_SIZE_KiB_mean_per_launch', 0.0) * 1024
        f_b = d.get('FETCH_SIZE_KiB_mean_per_launch', 0.0) * 1024
        # gfx950: FETCH_SIZE reports half the bytes of a coalesced stream (MI355X_MICROARCH.md, HBM) -> doubled
        d['hbm_bytes_per_launch'] = w_b + 2 * f_b
    rec = {'command': 'rocprofv3 --pmc <WRITE_SIZE|FETCH_SIZE> --output-format csv -- python3 bench.py --steps 5 --warmup 1 '
                      '--no-cpu-baseline   (one counter per pass)',
           'workload': 'stress: 4096 envs x 512 links, obs linear', 'kernels': kernels}
    (ROOT / 'profiles' / f'{tag}_pmc_hbm_traffic.json').write_text(json.dumps(rec, indent=1))
    print(out.read_text()[:600])
    for k, d in kernels.items():
        print(k[:60], {a: round(b) for a, b in d.items()})


if __name__ == '__main__':
    main()
