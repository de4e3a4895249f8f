"""Loader for the golden fixtures in tests/golden (data captured from the reference; see make_golden.py)."""
import json
from pathlib import Path
from types import SimpleNamespace

import numpy as np

GOLDEN_DIR = Path(__file__).resolve().parent / 'golden'


def case_names():
    return sorted(p.stem for p in GOLDEN_DIR.glob('case*.npz'))


def load_case(name):
    z = np.load(GOLDEN_DIR / f'{name}.npz')
    meta = json.loads(bytes(z['meta_json']).decode())
    ids = meta['dev_ids']
    index = {d: k for k, d in enumerate(ids)}
    steps = []
    for k in range(meta['num_steps']):
        keys = meta[f's{k}_keys']
        rec = {f[len(f's{k}_'):]: z[f] for f in z.files if f.startswith(f's{k}_')}
        rec['keys'] = keys
        rec['link_tx'] = np.asarray([index[key.split(':')[0]] for key in keys], dtype=np.int64)
        rec['link_rx'] = np.asarray([index[key.split(':')[1]] for key in keys], dtype=np.int64)
        steps.append(SimpleNamespace(**rec))
    return SimpleNamespace(name=name, meta=meta, ids=ids, cfgs=meta['dev_cfgs'], is_bs=z['dev_is_bs'],
                           pos=z['dev_pos'], steps=steps)


def known_answers():
    return json.loads((GOLDEN_DIR / 'known_answers.json').read_text())


def rel_err(got, ref):
    """max |d| / max(|ref|, 1) - the parity metric of BASELINE.md section 4."""
    got = np.asarray(got, dtype=np.float64); ref = np.asarray(ref, dtype=np.float64)
    if got.size == 0:
        return 0.0
    return float(np.max(np.abs(got - ref) / np.maximum(np.abs(ref), 1.0)))
