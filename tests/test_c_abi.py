"""The boundary from plain C: the header compiles as C99 and C++17, a C client links against the shared library
(CPU), and - on the GPU box - runs and agrees with the oracle."""
import json
import shutil
import subprocess
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent
LIB_DIR = ROOT / 'gym_d2d_amd' / 'lib'
SRC = ROOT / 'tests' / 'c' / 'abi_smoke.c'


def _build(tmp_path):
    from gym_d2d_amd import _native
    _native.load_library()              # makes sure the .so exists
    exe = tmp_path / 'abi_smoke'
    cmd = ['gcc', '-std=c99', '-Wall', '-Werror', '-I', str(ROOT / 'include'), str(SRC), '-L', str(LIB_DIR), '-ld2d_hip',
           f'-Wl,-rpath,{LIB_DIR}', '-lm', '-o', str(exe)]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return exe


def test_header_is_valid_c_and_cpp():
    for compiler, std in (('gcc', '-std=c99'), ('g++', '-std=c++17')):
        if shutil.which(compiler) is None:
            pytest.skip(f'{compiler} missing')
        for header in ('d2d_hip.h', 'd2d_hip_diag.h'):
            r = subprocess.run([compiler, std, '-Wall', '-Werror', '-pedantic', '-fsyntax-only', '-x',
                                'c' if compiler == 'gcc' else 'c++', str(ROOT / 'include' / header)],
                               capture_output=True, text=True)
            assert r.returncode == 0, r.stderr


def test_c_client_links(tmp_path):
    assert _build(tmp_path).exists()


@pytest.mark.gpu
def test_c_client_runs_and_matches_oracle(tmp_path):
    from oracle import d2d_oracle as orc
    exe = _build(tmp_path)
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    got = json.loads(r.stdout.strip().splitlines()[-1])
    ids, cfgs, is_bs = orc.device_configs(2, 2)
    cols = orc.device_columns(cfgs, is_bs)
    pos = np.array([[[0, 0], [100, 50], [-200, 120], [50, -80], [55, -70], [-300, 10], [-310, 25]]], dtype=np.float64)
    tx, rx, ty = np.array([1, 2, 3, 5]), np.array([0, 0, 4, 6]), np.array([1, 1, 3, 3])
    raw = np.array([[0 * 24 + 23, 1 * 24 + 10, 0 * 21 + 20, 1 * 21 + 5]])
    ref = orc.full_step(pos, tx, rx, ty, raw, cols, orc.PathLossSpec())
    assert got['flags'] == 0 and got['host_match'] == 1 and got['fixed_match'] == 1
    assert np.allclose(got['sinr_db'], ref['sinr_db'][0], rtol=1e-5, atol=1e-5)
    assert abs(got['reward'] - ref['reward'][0]) <= 1e-5 * max(1.0, abs(ref['reward'][0]))
    assert abs(got['obs_1_0'] - ref['obs'][0, 1, 0]) <= 1e-5 * max(1.0, abs(ref['obs'][0, 1, 0]))


@pytest.mark.gpu
def test_argument_validation_with_a_live_handle(tmp_path):
    """tests/c/abi_validation.c against the real library on the GPU box: its second half (call order, ranges, the diagnostic
    tuning keys a release build refuses) needs a live handle, which the CPU sanitizer pass never gets."""
    from gym_d2d_amd import _native
    _native.load_library()
    exe = tmp_path / 'abi_validation'
    r = subprocess.run(['gcc', '-std=c99', '-Wall', '-Werror', '-I', str(ROOT / 'include'), str(ROOT / 'tests' / 'c' / 'abi_validation.c'),
                        '-L', str(LIB_DIR), '-ld2d_hip', f'-Wl,-rpath,{LIB_DIR}', '-o', str(exe)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, (r.stdout, r.stderr)
    out = json.loads(r.stdout.strip().splitlines()[-1])
    assert out == {'gpu': 1, 'failures': 0}


def test_release_library_exports_exactly_the_product_header():
    """Every symbol include/d2d_hip.h declares is exported, and nothing else with the d2d_ prefix: the measurement equipment
    (write probes, phase stamps) lives in libd2d_probe.so / diagnostic builds (include/d2d_hip_diag.h)."""
    import re
    from gym_d2d_amd import _native
    _native.load_library()
    header = (ROOT / 'include' / 'd2d_hip.h').read_text()
    declared = set(re.findall(r'^(?:int|const char\*) (d2d_\w+)\(', header, flags=re.M))
    nm = subprocess.run(['nm', '-D', '--defined-only', str(LIB_DIR / 'libd2d_hip.so')], capture_output=True, text=True, check=True).stdout
    exported = {ln.split()[-1] for ln in nm.splitlines() if ' T d2d_' in ln}
    assert exported == declared, (sorted(exported - declared), sorted(declared - exported))
    assert set(_native.SIGNATURES) == declared
    assert len(declared) == 43          # ABI 5: the 41 of ABI 4 + d2d_set_positions_f64 + d2d_set_path_loss_link_table_dev
    probe = subprocess.run(['nm', '-D', '--defined-only', str(LIB_DIR / 'libd2d_probe.so')], capture_output=True, text=True, check=True).stdout
    assert {ln.split()[-1] for ln in probe.splitlines() if ' T d2d_' in ln} == {'d2d_probe_write_variants', 'd2d_probe_write_staged', 'd2d_probe_last_error'}
