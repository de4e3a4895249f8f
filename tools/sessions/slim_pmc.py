#!/usr/bin/env python3
"""Reduce one rocprofv3 --pmc output directory to the d2d kernels' counters: per kernel and counter the mean over dispatches of
the summed value and, where the JSON carries them, the per-instance (dimension) values; then delete the bulky raw files."""
import csv
import glob
import json
import os
import sys
import collections

root = sys.argv[1]
out = {'dir': os.path.basename(root), 'kernels': {}}
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(root + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'd2d::' in r['Kernel_Name']:
            acc[r['Kernel_Name'][:70]][r['Counter_Name']].append(float(r['Counter_Value']))
for k, d in acc.items():
    out['kernels'][k] = {c: {'mean': sum(v) / len(v), 'n': len(v), 'min': min(v), 'max': max(v)} for c, v in d.items()}
# per-instance values from the JSON (records[].counter_id + dimension ids), if present
for f in glob.glob(root + '/**/*results.json', recursive=True):
    try:
        j = json.load(open(f))
        tool = j['rocprofiler-sdk-tool'][0]
        names = {}
        for c in tool.get('counters', []):
            names[c['id']['handle'] if isinstance(c['id'], dict) else c['id']] = c['name']
        ksym = {k['kernel_id']: k.get('formatted_kernel_name', k.get('kernel_name', '')) for k in tool.get('kernel_symbols', [])}
        inst = collections.defaultdict(lambda: collections.defaultdict(list))
        for rec in tool.get('callback_records', {}).get('counter_collection', []):
            kid = rec['dispatch_data']['dispatch_info']['kernel_id']
            kn = ksym.get(kid, '')
            if 'd2d::' not in kn:
                continue
            per = collections.defaultdict(list)
            for r in rec['records']:
                cid = r['counter_id']['handle'] if isinstance(r['counter_id'], dict) else r['counter_id']
                per[names.get(cid, str(cid))].append(r['value'])
            for c, v in per.items():
                inst[kn[:70]][c].append(v)
        for kn, d in inst.items():
            for c, lists in d.items():
                n = min(len(x) for x in lists)
                if n > 1:
                    mean = [sum(x[i] for x in lists) / len(lists) for i in range(n)]
                    out['kernels'].setdefault(kn, {}).setdefault(c, {})['per_instance_mean'] = [round(x, 1) for x in mean]
    except Exception as e:           # noqa: BLE001
        out.setdefault('json_errors', []).append(f'{os.path.basename(f)}: {e!r}')
    os.remove(f)
print(json.dumps(out))
