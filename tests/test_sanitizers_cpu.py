"""Host-side sanitizer pass, CPU only - GPU AddressSanitizer is not available on the pool and nothing
here touches a GPU:

  * the C restatement of the oracle (oracle/c/d2d_oracle.c) rebuilt with gcc -fsanitize=address,undefined and driven
    through its usual ctypes front end over edge shapes (one link, everyone on one RB, out-of-range RBs, negative actions,
    OpenMP threads), results compared with the NumPy oracle;
  * the C-ABI layer's HOST code (csrc/d2d_capi.hip: argument validation, table building) rebuilt with
    hipcc -fsanitize=address,undefined (host instrumentation only) and called by tests/c/abi_validation.c - every entry
    point with null handles and out-of-range arguments - built with the same clang and sanitizers.
"""
import json
import os
import shutil
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent

ORACLE_DRIVER = r'''
import sys
from pathlib import Path
import numpy as np
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[1] + '/tests')
from oracle import c_oracle, d2d_oracle as orc
c_oracle.LIB_PATH = Path(sys.argv[2])
from sim_util import default_links, random_layout
worst = 0.0
rng = np.random.default_rng(5)
for b, rbs, cues, dues, threads in ((1, 1, 1, 0, 1), (3, 1, 5, 6, 1), (4, 7, 3, 0, 2), (2, 5, 0, 9, 1), (5, 25, 25, 25, 3), (2, 3, 40, 41, 2)):
    n = cues + dues
    pos = random_layout(rng, b, cues, dues).astype(np.float64)
    tx, rx, ty = default_links(cues, dues)
    cols = orc.device_columns(*orc.device_configs(cues, dues)[1:])
    raw = np.concatenate([rng.integers(0, rbs * 24, (b, cues)), rng.integers(0, rbs * 21, (b, dues))], 1)
    raw[0, 0] = -7                       # negative action: Python floor semantics (d2d_env.py:94-96)
    raw[-1, -1] = rbs * 24 * 3 + 1       # rb outside [0, R): accepted, like the reference
    for min_cap in (0.0, 1e9):
        want = orc.full_step(pos, tx, rx, ty, raw, cols, orc.PathLossSpec(), min_capacity_mbps=min_cap)
        got = c_oracle.full_step(pos, tx, rx, ty, raw, cols, orc.PathLossSpec(), min_capacity_mbps=min_cap, threads=threads)
        for k in ('sinr_db', 'snr_db', 'rate_bps', 'capacity_mbps', 'reward', 'table', 'obs'):
            w, g = np.asarray(want[k], dtype=np.float64), np.asarray(got[k], dtype=np.float64)
            assert w.shape == g.shape, (k, w.shape, g.shape)
            worst = max(worst, float(np.max(np.abs(w - g) / np.maximum(np.abs(w), 1.0))))
        assert np.array_equal(want['rb'], got['rb']) and np.array_equal(want['pwr'], got['pwr'])
print('WORST', worst)
assert worst <= 1e-12, worst
'''


def test_c_oracle_under_address_and_undefined_sanitizers(tmp_path):
    if shutil.which('gcc') is None:
        pytest.skip('gcc missing')
    libasan = subprocess.run(['gcc', '-print-file-name=libasan.so'], capture_output=True, text=True).stdout.strip()
    if not libasan or not Path(libasan).exists():
        pytest.skip('libasan missing')
    so = tmp_path / 'libd2d_oracle_c_san.so'
    r = subprocess.run(['gcc', '-O1', '-g', '-fopenmp', '-shared', '-fPIC', '-Wall', '-fsanitize=address,undefined',
                        '-fno-sanitize-recover=all', str(ROOT / 'oracle' / 'c' / 'd2d_oracle.c'), '-o', str(so), '-lm'],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    script = tmp_path / 'drive.py'
    script.write_text(ORACLE_DRIVER)
    env = dict(os.environ, LD_PRELOAD=libasan, ASAN_OPTIONS='detect_leaks=0:abort_on_error=1', UBSAN_OPTIONS='halt_on_error=1:print_stacktrace=1')
    r = subprocess.run([sys.executable, str(script), str(ROOT), str(so)], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    assert 'WORST' in r.stdout and 'ERROR: AddressSanitizer' not in r.stderr and 'runtime error' not in r.stderr


def test_c_abi_argument_validation_under_host_sanitizers(tmp_path):
    """No GPU is needed (and none is used here): every call is refused before any HIP work, d2d_create itself fails with
    D2D_ERR_HIP in this container.  What is checked is that the refusals are clean under ASan + UBSan, and - with the
    instrumented library built with -DD2D_TEST_HOOKS=1 - that a std::bad_alloc, a std::exception and a foreign exception
    thrown inside an entry point come back as D2D_ERR_NO_MEMORY / D2D_ERR_STATE with a message (the exception barrier)."""
    from gym_d2d_amd import build as b
    hipcc = b._hipcc()
    clang = Path('/opt/rocm/lib/llvm/bin/clang')
    if not clang.exists():
        pytest.skip('ROCm clang missing')
    b.build()                                                   # the kernels' objects (not instrumented) are reused
    obj_dir = b.LIB_DIR / 'obj'
    san = ['-fsanitize=address,undefined', '-fno-sanitize-recover=undefined']
    capi = tmp_path / 'd2d_capi_san.o'
    flags = [f for f in b.FLAGS if f != '-O3']
    r = subprocess.run([hipcc, '-O1', '-g', *flags, *san, '-DD2D_TEST_HOOKS=1', '-I', str(b.INCLUDE), '-c', str(b.CSRC / 'd2d_capi.hip'), '-o', str(capi)],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    lib = tmp_path / 'libd2d_hip.so'
    objs = [str(obj_dir / f'{Path(s).stem}.o') for s in b.SOURCES if s != 'd2d_capi.hip']
    r = subprocess.run([hipcc, '-shared', '-fPIC', f'--offload-arch={b.ARCH}', *san, '-o', str(lib), str(capi), *objs],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    exe = tmp_path / 'abi_validation'
    r = subprocess.run([str(clang), '-std=c99', '-O1', '-g', '-Wall', '-Werror', *san, '-DD2D_TEST_HOOKS=1', '-I', str(ROOT / 'include'),
                        str(ROOT / 'tests' / 'c' / 'abi_validation.c'), '-L', str(tmp_path), '-ld2d_hip', f'-Wl,-rpath,{tmp_path}',
                        '-o', str(exe)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    env = dict(os.environ, ASAN_OPTIONS='detect_leaks=0:abort_on_error=1', UBSAN_OPTIONS='halt_on_error=1:print_stacktrace=1')
    env.pop('LD_PRELOAD', None)
    r = subprocess.run([str(exe)], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    out = json.loads(r.stdout.strip().splitlines()[-1])
    assert out['failures'] == 0
    assert 'AddressSanitizer' not in r.stderr and 'runtime error' not in r.stderr
