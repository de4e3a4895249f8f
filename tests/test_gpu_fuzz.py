"""Short runs of the fuzzers under tools/ inside the GPU suite (a few seconds each, fresh random seeds every run): step-kernel
variants bit-identical to the all-pairs sweep, the reset sampler against the oracle, the dict-style D2DEnv and the batched
VecD2DEnv against the oracle over random sizes / link subsets / traffic models / rewards.  The long runs that found round 3's
numerics problems are the same scripts with a larger time budget."""
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent


@pytest.mark.parametrize('script,seconds,token', [('fuzz_variants.py', 10, 'fuzz ok'), ('fuzz_reset.py', 5, 'reset fuzz ok'),
                                                  ('fuzz_dropin.py', 6, 'drop-in fuzz ok'), ('fuzz_vec_env.py', 8, 'vec env fuzz ok')])
def test_fuzzer_runs_clean(script, seconds, token):
    r = subprocess.run([sys.executable, str(ROOT / 'tools' / script), str(seconds)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and token in r.stdout, (r.stdout[-1500:], r.stderr[-1500:])
