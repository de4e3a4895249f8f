"""Build libd2d_hip.so (the product: the C ABI of include/d2d_hip.h) and libd2d_probe.so (measurement equipment: the
write-ceiling probe of include/d2d_hip_diag.h) for gfx950 with hipcc - in-tree, so the .so files travel with the repo snapshot.

    python -m gym_d2d_amd.build [--force] [--verbose]
    D2D_BUILD_DIAG=1 python -m gym_d2d_amd.build      # diagnostic build of libd2d_hip.so: the A/B tuning keys and the ablation
                                                      # switch of include/d2d_hip_diag.h are accepted (tools/ab_step.py ...)
"""
from __future__ import annotations

import hashlib
import os
import shutil
import subprocess
import sys
from pathlib import Path

PKG = Path(__file__).resolve().parent
CSRC = PKG / 'csrc'
LIB_DIR = PKG / 'lib'
LIB_PATH = LIB_DIR / 'libd2d_hip.so'
PROBE_PATH = LIB_DIR / 'libd2d_probe.so'
INCLUDE = PKG.parent / 'include'
ARCH = 'gfx950'

SOURCES = ['d2d_step.hip', 'd2d_rollout.hip', 'd2d_obs.hip', 'd2d_reset.hip', 'd2d_gain.hip', 'd2d_capi.hip']
PROBE_SOURCES = ['d2d_probe.hip']
HEADERS = [CSRC / 'd2d_internal.h', CSRC / 'd2d_step_device.h', CSRC / 'd2d_store.h', INCLUDE / 'd2d_hip.h', INCLUDE / 'd2d_hip_diag.h']
FLAGS = ['-O3', '-std=c++17', '-fPIC', f'--offload-arch={ARCH}', '-fno-gpu-rdc', '-Wall', '-Wno-unused-function', '-Wno-unused-value',
         # the kernels already issue their uniform-address LDS atomics from one lane (or on rare paths): LLVM's atomic optimizer
         # only wraps them in mbcnt / readlane / popcount-multiply sequences
         '-mllvm', '-amdgpu-atomic-optimizer-strategy=None']
FLAGS += os.environ.get('D2D_BUILD_DEFINES', '').split()       # experiment builds (tools/ab_builds.py), e.g. -DD2D_EXP_PF_POS=1
if os.environ.get('D2D_BUILD_DIAG') == '1':          # diagnostic build: the step kernel honours D2D_TUNE_STEP_ABLATE
    FLAGS += ['-DD2D_DIAG=1', '-DD2D_STEP_ABLATE=1']


def _hipcc() -> str:
    for cand in (os.environ.get('HIPCC'), shutil.which('hipcc'), '/opt/rocm/bin/hipcc'):
        if cand and Path(cand).exists():
            return cand
    raise RuntimeError('hipcc not found: libd2d_hip.so cannot be built (there is no CPU fallback)')


def source_digest() -> str:
    """sha256 over the kernel / C-ABI sources, headers and compile flags: identifies what a profile was taken on."""
    h = hashlib.sha256()
    for p in [CSRC / s for s in SOURCES + PROBE_SOURCES if (CSRC / s).exists()] + [h for h in HEADERS if h.exists()]:
        h.update(p.name.encode()); h.update(p.read_bytes())
    h.update(' '.join(FLAGS).encode())
    return h.hexdigest()


def build(force: bool = False, verbose: bool = False) -> Path:
    LIB_DIR.mkdir(exist_ok=True)
    stamp = LIB_DIR / 'libd2d_hip.sha256'
    digest = source_digest()
    if not force and LIB_PATH.exists() and stamp.exists() and stamp.read_text().strip() == digest:
        return LIB_PATH
    hipcc = _hipcc()
    obj_dir = LIB_DIR / 'obj'
    obj_dir.mkdir(exist_ok=True)
    procs = []
    for s in SOURCES + PROBE_SOURCES:
        src = CSRC / s
        obj = obj_dir / (src.stem + '.o')
        cmd = [hipcc, *FLAGS, '-I', str(INCLUDE), '-c', str(src), '-o', str(obj)]
        if verbose:
            print(' '.join(cmd), flush=True)
        procs.append((s, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    for s, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f'hipcc failed on {s}:\n{out}')
        if verbose and out.strip():
            print(out)
    for lib, sources in ((LIB_PATH, SOURCES), (PROBE_PATH, PROBE_SOURCES)):
        objs = [str(obj_dir / (Path(s).stem + '.o')) for s in sources]
        cmd = [hipcc, '-shared', '-fPIC', f'--offload-arch={ARCH}', '-o', str(lib), *objs]
        if verbose:
            print(' '.join(cmd), flush=True)
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if r.returncode != 0:
            raise RuntimeError(f'link failed:\n{r.stdout}')
    stamp.write_text(digest)
    return LIB_PATH


if __name__ == '__main__':
    path = build(force='--force' in sys.argv, verbose='--verbose' in sys.argv or '-v' in sys.argv)
    print(path)
