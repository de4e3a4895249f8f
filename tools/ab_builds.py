#!/usr/bin/env python3
"""Same-box A/B of library BUILDS (box-to-box spread on the pool is +-5 %, more than most kernel changes are worth).

    python tools/ab_builds.py prepare <commit|WORKTREE> ...     # here (no GPU): build each into tools/ab_libs/<name>.so
    python tools/ab_builds.py run [--workload stress|default] [--passes 2]      # on the GPU box: alternate processes, one per build

`prepare` uses git worktrees for commits and the working tree's own build for WORKTREE; the .so files are git-ignored but
travel with the gpurun snapshot.  `run` times each build in its own process (the library path is patched before the
first load), alternating builds so that drift hits all of them alike, and prints one JSON line per (pass, build).
"""
import argparse
import json
import shutil
import statistics
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
LIBS = ROOT / 'tools' / 'ab_libs'


def prepare(names):
    LIBS.mkdir(exist_ok=True)
    for old in LIBS.glob('*.so'):
        old.unlink()
    for k, name in enumerate(names):
        out = LIBS / f'{k:02d}_{name.replace("/", "_")}.so'
        if name == 'WORKTREE':
            subprocess.run([sys.executable, '-m', 'gym_d2d_amd.build', '--force'], cwd=ROOT, check=True, stdout=subprocess.DEVNULL)
            shutil.copy(ROOT / 'gym_d2d_amd' / 'lib' / 'libd2d_hip.so', out)
        else:
            wt = Path('/tmp') / f'ab_wt_{k}'
            subprocess.run(['git', 'worktree', 'remove', '--force', str(wt)], cwd=ROOT, stderr=subprocess.DEVNULL)
            subprocess.run(['git', 'worktree', 'add', '-q', str(wt), name], cwd=ROOT, check=True)
            subprocess.run([sys.executable, '-m', 'gym_d2d_amd.build', '--force'], cwd=wt, check=True, stdout=subprocess.DEVNULL)
            shutil.copy(wt / 'gym_d2d_amd' / 'lib' / 'libd2d_hip.so', out)
            subprocess.run(['git', 'worktree', 'remove', '--force', str(wt)], cwd=ROOT)
        print('built', out.name)


def one(lib, workload, export=True, mode='table', walk=-1, lpt=-1, path_loss='log2', positions='float32'):
    sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / 'tools'))
    from gym_d2d_amd import _native
    _native.LIB_PATH = Path(lib).resolve()
    # an older build lacks the newer entry points: drop them from the binding (this tool only) and make their wrappers no-ops
    import ctypes
    probe = ctypes.CDLL(str(_native.LIB_PATH))
    _native.ABI_VERSION = probe.d2d_abi_version()            # an older build: its own ABI number (the calls this tool makes exist in all)
    for name in [n for n in _native.SIGNATURES if not hasattr(probe, n)]:
        del _native.SIGNATURES[name]
        setattr(_native.Handle, name[len('d2d_'):], lambda self, *a, **k: None)
    import torch
    from ab_step import timed
    from gym_d2d_amd.envs import VecD2DEnv
    from gym_d2d_amd.envs.obs_fn import LinearObsFunction, OwnLinkObsFunction, SignalPlanesObsFunction
    extra = {}
    if path_loss == 'cost_hata':                 # an exponent other than 2: the power-law kernels
        from gym_d2d_amd.path_loss import CostHataPathLoss
        extra = {'path_loss_model': CostHataPathLoss}
    if workload == 'stress':
        b, c, p, r = 4096, 256, 256, 256
        if mode == 'none':           # the learner configuration: no table, SystemCapacity's scalar once per env
            env = VecD2DEnv({'num_rbs': r, 'num_cues': c, 'num_due_pairs': p, 'obs_fn': SignalPlanesObsFunction, **extra}, num_envs=b, reward_per_env=True)
        else:
            env = VecD2DEnv({'num_rbs': r, 'num_cues': c, 'num_due_pairs': p, 'obs_fn': OwnLinkObsFunction, **extra}, num_envs=b)
        cols = c + p
    else:
        b, c, p, r = 1024, 25, 25, 25
        env = VecD2DEnv({'num_rbs': r, 'num_cues': c, 'num_due_pairs': p, 'obs_fn': LinearObsFunction}, num_envs=b, cue_actions='traffic')
        cols = p
    env.reset(seed=1)
    h = env.simulator.handle
    if positions == 'float64':                   # the layout the sampler drew, moved off the float32 grid: the (hi, lo) kernels
        import numpy as np
        pos = env.simulator.positions().astype(np.float64)
        pos[:, 1:] += np.random.default_rng(1).uniform(-1e-6, 1e-6, pos[:, 1:].shape)
        env.simulator.set_positions(pos)
    h.set_export_actions(export)
    if walk >= 0:
        h.set_tuning(_native.TUNE_STEP_WALK, walk)
    if lpt > 0:
        h.set_tuning(_native.TUNE_STEP_LPT, lpt)
    act = torch.randint(0, r * 21, (64, b, cols), device=env.device, dtype=torch.int32)
    t = [timed(h, act, 32) for _ in range(15)]
    print(json.dumps({'build': Path(lib).stem, 'workload': workload, 'path_loss': path_loss, 'positions': positions, 'obs': mode, 'walk': walk, 'lpt': lpt, 'export_rb_pwr': int(export), 'median_us': round(statistics.median(t), 2), 'min_us': round(min(t), 2)}))


def run(workload, passes, export=True, mode='table', walk=-1, lpt=-1, path_loss='log2', positions='float32'):
    for k in range(passes):
        for lib in sorted(LIBS.glob('*.so')):
            r = subprocess.run([sys.executable, __file__, 'one', str(lib), '--workload', workload, '--mode', mode, '--walk', str(walk), '--lpt', str(lpt), '--path-loss', path_loss, '--positions', positions] + ([] if export else ['--no-export']),
                               capture_output=True, text=True)
            line = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
            print(line[-1] if line else f'{lib.name}: failed {r.stderr[-300:]}', flush=True)


if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('what', choices=['prepare', 'run', 'one'])
    ap.add_argument('names', nargs='*')
    ap.add_argument('--workload', default='stress')
    ap.add_argument('--passes', type=int, default=2)
    ap.add_argument('--mode', default='table', choices=['table', 'none'], help='stress: compact table (OwnLinkObs) or the obs-less learner mode')
    ap.add_argument('--walk', type=int, default=-1, help='D2D_TUNE_STEP_WALK (-1 = the library default)')
    ap.add_argument('--lpt', type=int, default=-1, help='D2D_TUNE_STEP_LPT (-1 = the library default)')
    ap.add_argument('--path-loss', default='log2', choices=['log2', 'cost_hata'], help='log-distance with exponent 2 (1 / d^2 kernels) or COST-Hata (power-law kernels)')
    ap.add_argument('--positions', default='float32', choices=['float32', 'float64'], help='float64: host-uploaded layout off the float32 grid (d2d_set_positions_f64, OPT_XPOS kernels)')
    ap.add_argument('--no-export', action='store_true', help='d2d_set_export_actions(0): no decoded rb / pwr planes')
    a = ap.parse_args()
    if a.what == 'prepare':
        prepare(a.names)
    elif a.what == 'run':
        run(a.workload, a.passes, not a.no_export, a.mode, a.walk, a.lpt, a.path_loss, a.positions)
    else:
        one(a.names[0], a.workload, not a.no_export, a.mode, a.walk, a.lpt, a.path_loss, a.positions)
