#!/usr/bin/env python3
"""Benchmark of the hot path: D2DEnv.step for a batch of environments on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload stress|default|plugin|hata] [--obs linear|table|none]

A "step" is one pass of the fused path over the whole batch: action decode -> SINR / SNR / rate / capacity -> reward
-> observation table -> LinearObs expansion (reference semantics: obs materialised as [B, N, 6N] in HBM), with fresh
i.i.d. actions every step (pre-generated in HBM; positions fixed for the run - the reference's step never moves
devices, simulator.py:61-75), driven through the PUBLIC batched API: VecD2DEnv.step(actions) -> (obs, rewards, dones, info).
Metric: agent-steps/s = B * N * steps / wall seconds (whole job, all GPUs).

N > 1: one process per GPU (torch.distributed / RCCL), env axis sharded 4096 per GPU (weak scaling), one all-gather
per step of rewards + the per-step columns of the compact obs table, overlapped on a side stream (--gather rewards and
--signal-every K select the lighter plans).  `--gpus N` works both ways: under `python -m torch.distributed.run
--nproc-per-node N bench.py --gpus N` (RANK is already set: this process IS a rank) and as a plain `python bench.py
--gpus N` - then this process is only a launcher: it starts the N rank processes BEFORE anything touches HIP (it never
imports torch, never re-execs), waits, and exits non-zero if any rank failed.

Prints ONE JSON line on rank 0 with `roofline` (dominant kernel, HIP-event timed on the library's stream) and
`cpu_baseline` (the oracle's plain-C restatement timed on this box's host cores on a bounded sample; rank 0, N = 1 only;
one thread, with the all-core figure and the NumPy oracle's beside it).  Beside `value` and never as it:
  N = 1  `core_mode` (the same envs stepped with the compact-table obs: throughput + the step kernel's roofline block,
         SURVEY.md 8(d); without and with the optional decoded (rb, pwr) planes), `other_workloads` (BASELINE configs 2 and 4
         and an episode loop with the device-side reset, each with its own roofline block), `vec_env_step_ms` (the public
         VecD2DEnv.step against the bare C-ABI handle loop), `box_write_ceiling` / `roofline.box_ceiling_GBs` (the best of a
         family of pure fill kernels on this box), `single_env_step_ms` (the drop-in single-env D2DEnv.step, host dicts in /
         out); `--no-extras` keeps the headline only;
  N > 1  `rccl_ranks` + all-reduce / all-gather checksums, `value_per_gpu`, `per_rank` (every rank's own ms per step and kernel
         times), `gather` (bytes per GPU and step, side-stream ms per step, the same steps without the gather, the exposed
         difference), and `core_mode` with the rewards-only gather plan.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

import numpy as np

WORKLOADS = {
    # BASELINE.json configs[2]: the configuration the target metric is quoted on
    'stress': dict(name='4096 envs x (256 CUE + 256 DUE pairs, 256 RB), LogDistance', envs=4096, rbs=256, cues=256, dues=256),
    # BASELINE.json configs[1]
    'default': dict(name='1024 envs x (25 CUE + 25 DUE pairs, 25 RB), LogDistance, UplinkTrafficModel-driven CUEs',
                    envs=1024, rbs=25, cues=25, dues=25, traffic=True),
    # BASELINE.json configs[3]: the plugin-ABI swap - FreeSpacePathLoss class through env_config['path_loss_model'] and
    # a custom array ObsFunction (own link only); same sizes as 'stress'
    'plugin': dict(name='4096 envs x (256 CUE + 256 DUE pairs, 256 RB), FreeSpacePathLoss + OwnLinkObsFunction plugins',
                   envs=4096, rbs=256, cues=256, dues=256, plugin=True),
    # rows a5 / a5'' at the stress sizes: an exponent other than 2 (COST-Hata, path_loss.py:90-123) -> the power-law kernels, which
    # no BASELINE configuration runs (the lowest HBM fraction in the repo had no evidence of its own)
    'hata': dict(name='4096 envs x (256 CUE + 256 DUE pairs, 256 RB), CostHataPathLoss (power-law kernel)',
                 envs=4096, rbs=256, cues=256, dues=256, hata=True),
}
HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8 TB/s spec


# ------------------------------------------------------------------------------------------------ CPU baseline
def _cpu_chunk(job):
    """One worker's share of the CPU baseline: oracle steps of `envs` envs each until `calls` calls are done or the
    wall-clock `deadline` (time.time()) has passed, whichever comes first (module-level: picklable)."""
    c, p, r, envs, calls, seed, deadline = job
    for var in ('OMP_NUM_THREADS', 'OPENBLAS_NUM_THREADS', 'MKL_NUM_THREADS'):
        os.environ[var] = '1'
    from oracle import d2d_oracle as orc
    sys.path.insert(0, str(ROOT / 'tests'))
    from sim_util import default_links, random_layout
    rng = np.random.default_rng(seed)
    ids, cfgs, is_bs = orc.device_configs(c, p)
    cols = orc.device_columns(cfgs, is_bs)
    tx, rx, ty = default_links(c, p)
    pos = random_layout(rng, envs, c, p).astype(np.float64)
    t0 = time.perf_counter()
    done = 0
    while done < calls and (done == 0 or time.time() < deadline):
        raw = np.concatenate([rng.integers(0, r * 24, (envs, c)), rng.integers(0, r * 21, (envs, p))], 1)
        orc.full_step(pos, tx, rx, ty, raw, cols, orc.PathLossSpec(), with_obs=True, chunk=envs)
        done += 1
    return envs * done, time.perf_counter() - t0


def effective_cores():
    """Cores this process can really use: its affinity mask, capped by the cgroup CPU quota (a container that sees 256
    cores may be entitled to 16 of them: more workers than that only time-share)."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = Path('/sys/fs/cgroup/cpu.max').read_text().split()[:2]
        if quota != 'max':
            n = min(n, max(1, int(float(quota) / float(period) + 0.5)))
    except Exception:
        try:
            q = int(Path('/sys/fs/cgroup/cpu/cpu.cfs_quota_us').read_text())
            per = int(Path('/sys/fs/cgroup/cpu/cpu.cfs_period_us').read_text())
            if q > 0:
                n = min(n, max(1, int(q / per + 0.5)))
        except Exception:
            pass
    return n


def _c_oracle_leg(w, seconds, threads):
    """The C restatement of the oracle (oracle/c/d2d_oracle.c) on the same workload, obs materialised, for `seconds`."""
    from oracle import c_oracle
    from oracle import d2d_oracle as orc
    sys.path.insert(0, str(ROOT / 'tests'))
    from sim_util import default_links, random_layout
    c, p, r = w['cues'], w['dues'], w['rbs']
    n = c + p
    envs = max(threads, 1) * (1 if n > 100 else 32)
    rng = np.random.default_rng(4321)
    ids, cfgs, is_bs = orc.device_configs(c, p)
    cols = orc.device_columns(cfgs, is_bs)
    tx, rx, ty = default_links(c, p)
    pos = random_layout(rng, envs, c, p).astype(np.float64)
    raws = [np.concatenate([rng.integers(0, r * 24, (envs, c)), rng.integers(0, r * 21, (envs, p))], 1) for _ in range(3)]
    out = c_oracle.full_step(pos, tx, rx, ty, raws[0], cols, orc.PathLossSpec(), threads=threads)      # allocates + warms up
    t0 = time.perf_counter()
    calls = 0
    while calls == 0 or time.perf_counter() - t0 < seconds:
        c_oracle.full_step(pos, tx, rx, ty, raws[calls % 3], cols, orc.PathLossSpec(), threads=threads, out=out)
        calls += 1
    dt = time.perf_counter() - t0
    return envs * calls * n / dt, envs * calls, dt


def cpu_baseline(w, seconds_budget=16.0):
    """The oracle - a port of the reference's algorithm; the Python reference cannot travel to this box - timed on a
    bounded sample of the same workload, LinearObs materialised (float64, as the reference's arrays are).  `value` is the
    plain-C restatement on one thread, `all_cores` the same with OpenMP over envs on every core this process may use;
    `numpy_oracle` holds the NumPy restatement's figures (one thread, and a process pool over every core).  Every leg is
    bounded by a wall-clock deadline.  Runs BEFORE this process initialises HIP, so forking workers is safe."""
    n = w['cues'] + w['dues']
    cores = effective_cores()
    out = {}
    try:
        v1, steps1, dt1 = _c_oracle_leg(w, 4.0, 1)
        out = {'value': v1, 'unit': 'agent-steps/s', 'cores': 1, 'kind': 'port',
               'sample': f'{steps1} env-steps of the same workload ({steps1 * n} agent-steps, obs materialised in float64), '
                         f'C restatement of the oracle (oracle/c/d2d_oracle.c, gcc -O2), one thread, {dt1:.1f} s; host has '
                         f'{os.cpu_count()} cores',
               'reference_pure_python_1core_build_container': 6.3e3 if n > 100 else 3.3e4}
        va, stepsa, dta = _c_oracle_leg(w, 4.0, cores)
        out['all_cores'] = {'value': va, 'unit': 'agent-steps/s', 'cores': cores,
                            'sample': f'{stepsa} env-steps, OpenMP over envs on {cores} threads (affinity mask capped by the '
                                      f'cgroup CPU quota; the host shows {os.cpu_count()} cores), {dta:.1f} s'}
    except Exception as exc:                      # pragma: no cover - a baseline failure must not lose the GPU number
        out['c_oracle_error'] = repr(exc)
    numpy_leg = _numpy_baseline(w, seconds_budget)
    if 'value' not in out:
        numpy_leg.update(out)
        return numpy_leg
    out['numpy_oracle'] = numpy_leg
    return out


def _numpy_baseline(w, seconds_budget):
    """The NumPy fp64 oracle: one thread, and a process pool over env chunks on every core."""
    c, p, r = w['cues'], w['dues'], w['rbs']
    n = c + p
    envs = 8 if n > 100 else 128
    done, dt = _cpu_chunk((c, p, r, envs, 1 << 30, 1234, time.time() + 0.3 * seconds_budget))
    single = done * n / dt
    out = {'value': single, 'unit': 'agent-steps/s', 'cores': 1, 'kind': 'port',
           'sample': f'{done} env-steps of the same workload ({done * n} agent-steps, obs materialised), '
                     f'NumPy fp64 oracle, single thread, {dt:.1f} s; host has {os.cpu_count()} cores'}
    # -- every core: one process per core (NumPy elementwise code holds the GIL; BLAS threads pinned to 1)
    try:
        import multiprocessing as mp
        cores = effective_cores()
        with mp.get_context('fork').Pool(cores) as pool:
            t0 = time.perf_counter()
            deadline = time.time() + 0.3 * seconds_budget
            res = pool.map(_cpu_chunk, [(c, p, r, envs, 1 << 30, 1000 + k, deadline) for k in range(cores)], chunksize=1)
            wall = time.perf_counter() - t0
        total = sum(d for d, _ in res)
        out['all_cores'] = {'value': total * n / wall, 'unit': 'agent-steps/s', 'cores': cores,
                            'sample': f'{total} env-steps over {cores} worker processes (one per core, 1 NumPy/BLAS thread '
                                      f'each, {envs} envs per oracle call, each worker runs until a shared deadline), '
                                      f'{wall:.1f} s wall'}
    except Exception as exc:                      # pragma: no cover - a baseline failure must not lose the GPU number
        out['all_cores'] = {'error': repr(exc)}
    return out


# ------------------------------------------------------------------------------------------------ launcher
def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def launch_ranks(n, argv, rank_timeout=0.0):
    """Start one rank process per GPU and wait for them.  This process never touches HIP (torch is not even imported
    here) and never execs: the ranks are ordinary children.  Rank 0 inherits stdout (the JSON line); the other ranks'
    stdout goes to stderr.  Returns the exit code (first failing rank's, else 0).  rank_timeout > 0: a deadline in seconds
    for the whole job - a rank stuck in a collective would otherwise hold the job until the driver's own limit; on expiry
    the launcher terminates ITS OWN children by PID (kill after a grace period) and returns 124."""
    port = os.environ.get('MASTER_PORT') or str(_free_port())
    deadline = time.monotonic() + rank_timeout if rank_timeout > 0 else None
    procs = []
    for rank in range(n):
        env = dict(os.environ)
        env.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR='127.0.0.1', MASTER_PORT=port, D2D_BENCH_LAUNCHED='1')
        env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        procs.append(subprocess.Popen([sys.executable, str(Path(__file__).resolve())] + list(argv), env=env,
                                      stdout=None if rank == 0 else sys.stderr))
    code = 0
    live = set(range(n))
    while live:
        for rank in sorted(live):
            rc = procs[rank].poll()
            if rc is None:
                continue
            live.discard(rank)
            if rc != 0 and code == 0:
                code = rc if rc > 0 else 1
                print(f'bench launcher: rank {rank} exited with {rc}; stopping the other ranks', file=sys.stderr)
                for other in live:
                    procs[other].terminate()            # exact PIDs of our own children
        if deadline is not None and live and time.monotonic() > deadline:
            print(f'bench launcher: ranks {sorted(live)} still running after --rank-timeout {rank_timeout:g} s; terminating them',
                  file=sys.stderr)
            for rank in live:
                procs[rank].terminate()
            grace = time.monotonic() + 5.0
            while any(procs[r].poll() is None for r in live) and time.monotonic() < grace:
                time.sleep(0.05)
            for rank in live:
                if procs[rank].poll() is None:
                    procs[rank].kill()
                procs[rank].wait()
            return 124
        time.sleep(0.05)
    return code


# ------------------------------------------------------------------------------------------------ test stub
class _StubHandle:
    """CPU stand-in for the native handle (launcher / gather plumbing test on gloo; never a measurement)."""

    def __init__(self, env):
        self.env = env

    def step(self, ptr=0):
        self.env.k += 1
        self.env._t['reward'].fill_(float(self.env.first_env + self.env.k))
        self.env._t['table'][:, :, 4:] = float(self.env.k)

    def reset_positions(self, *a):
        pass

    def profile_reset(self):
        pass

    def profile_enable(self, on):
        pass

    def profile_read(self, k):
        return 0.0, 0

    def set_obs_mode(self, m):
        pass

    def set_export_actions(self, on):
        pass


class _StubEnv:
    def __init__(self, torch, b, n, first_env):
        self.k, self.first_env = 0, first_env
        self._t = {'reward': torch.zeros(b, n), 'table': torch.zeros(b, n, 6)}
        self._t['table'][:, :, :4] = torch.arange(first_env, first_env + b, dtype=torch.float32)[:, None, None]
        self.num_pwr_actions = {'cue': 24, 'due': 21, 'mbs': 47}
        self.simulator = type('S', (), {})()
        self.simulator.handle = _StubHandle(self)

    def reset(self, seed=None):
        pass

    def status_flags(self):
        return 0

    def close(self):
        pass


# ------------------------------------------------------------------------------------------------ one rank
def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=100)
    ap.add_argument('--warmup', type=int, default=10)
    ap.add_argument('--workload', default='stress', choices=sorted(WORKLOADS))
    ap.add_argument('--obs', default='linear', choices=['linear', 'table', 'none'])
    ap.add_argument('--obs-dtype', default='float32', choices=['float32', 'float64'],
                    help="--obs linear: element type of the [B, N, 6N] block; float64 is the reference's own (obs_fn.py:47,51), written by the "
                         'expansion kernel itself (48 N bytes per agent-step)')
    ap.add_argument('--envs', type=int, default=0, help='override envs per GPU')
    ap.add_argument('--cue-actions', default='', choices=['', 'agent', 'traffic'],
                    help="who drives the CUE links (default: the workload's own choice)")
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-gather', action='store_true', help='N > 1: skip the per-step all-gather')
    ap.add_argument('--gather', default='table', choices=['table', 'rewards', 'planes'],
                    help="N > 1: what the per-step all-gather carries (StepGatherer mode): rewards + the (sinr, snr) columns of the "
                         "obs table, or rewards only")
    ap.add_argument('--gather-backend', default='torch', choices=['torch', 'native'],
                    help="N > 1: torch.distributed collectives (nccl = RCCL) or the library's own RCCL entry (d2d_comm_init / d2d_allgather)")
    ap.add_argument('--reward-every', type=int, default=8,
                    help='N > 1, rewards-only gather plan (--gather rewards, and core_mode): the rewards of K steps ride in a device-side '
                         'ring and travel together (one gather launch costs more host time than a compact-obs step takes)')
    ap.add_argument('--signal-every', type=int, default=1, help='N > 1, --gather table: the (sinr, snr) columns travel on every K-th step')
    ap.add_argument('--with-reset', action='store_true', help='redraw all device positions every 10 steps (device-side reset)')
    ap.add_argument('--force-dist', action='store_true', help='init RCCL and run the gather path even with one rank (test hook)')
    ap.add_argument('--no-single-env-latency', action='store_true',
                    help='N = 1 also times the drop-in single-env D2DEnv.step (host dicts in / out); this skips it')
    ap.add_argument('--rank-timeout', type=float, default=1500.0,
                    help='N > 1, self-launched ranks: seconds after which the launcher terminates its children and exits 124 '
                         '(0 = no deadline)')
    ap.add_argument('--no-extras', action='store_true',
                    help='N = 1, default run: skip other_workloads / vec_env_step_ms / the write-ceiling probe (the headline only)')
    ap.add_argument('--no-export', action='store_true',
                    help='d2d_set_export_actions(0) for the measured session: no decoded (rb, pwr) planes (a rollout knows its actions)')
    ap.add_argument('--reward-per-env', action='store_true',
                    help="--obs table / none: SystemCapacity's reward once per env (VecD2DEnv(reward_per_env=True), D2D_REWARD_PER_ENV) "
                         'instead of the [B, N] copy')
    ap.add_argument('--tune', default='', help='comma list key=value: rows,nt,xcd,bucket,block,threads,epw,sblock,fuse,walk')
    ap.add_argument('--stub-cpu', action='store_true',
                    help='TEST HOOK: no GPU, gloo backend, synthetic per-rank results - exercises only the launcher / gather plumbing')
    ap.add_argument('--share-gpu', action='store_true',
                    help='TEST HOOK: every rank uses cuda:0 and gloo carries the collectives (RCCL refuses two ranks on one '
                         'GPU) - the real worker with N > 1 on a one-GPU box; never a measurement')
    return ap.parse_args(argv)


GROUP = 20          # launches per HIP-event pair when a step is a single kernel
CORE_BYTES = 40.0   # action 4 + positions 16 + outputs 16 + reward 4, per agent-step


def algorithmic_bytes(n, obs, reward_per_env=False, export=False, f64=False):
    """Per agent-step: the 40 core bytes + 24 N for the materialised LinearObs (48 N as float64; + the 24 the
    expansion reads) or + 24 for the compact table.  reward_per_env: SystemCapacity's scalar once per env (4 / N bytes per link)
    instead of the 4-byte copy every agent gets.  export: + 8 for the decoded (rb, pwr) planes behind info['rb'] /
    info['tx_pwr_dbm'] when the timed configuration writes them (d2d_set_export_actions(1), the public default)."""
    core = CORE_BYTES - 4.0 + 4.0 / n if reward_per_env else CORE_BYTES
    return core + (8.0 if export else 0.0) + ((48.0 if f64 else 24.0) * n if obs == 'linear' else (24.0 if obs == 'table' else 0.0))


class Session:
    """One workload on this rank's GPU: the env, its pre-generated actions and the timing loops."""

    def __init__(self, torch, args, key, obs, dev, rank, local, steps, warmup, *, envs=0, cue_mode='', tune='', stub=False,
                 export=True, action_pool=0, placement_trials=0, f64=False, reward_per_env=None):
        self.torch, self.args, self.key, self.obs, self.dev, self.rank, self.stub = torch, args, key, obs, dev, rank, stub
        self.export = export
        self.f64 = bool(f64) and obs == 'linear'
        w = dict(WORKLOADS[key])
        if envs:
            w['envs'] = envs
        self.w = w
        b, c, p, r = w['envs'], w['cues'], w['dues'], w['rbs']
        self.b, self.c, self.p, self.r, self.n = b, c, p, r, c + p
        if w.get('plugin') and obs == 'linear':
            self.obs = obs = 'table'
        self.cue_mode = cue_mode or ('traffic' if w.get('traffic') else 'agent')
        self.steps, self.warmup, self.total = steps, warmup, steps + warmup
        if stub:
            self.env = _StubEnv(torch, b, self.n, rank * b)
            self.h = self.env.simulator.handle
            self.n_agents = self.n
        else:
            from gym_d2d_amd import _native
            from gym_d2d_amd.envs import VecD2DEnv
            from gym_d2d_amd.envs.obs_fn import LinearObsFunction, OwnLinkObsFunction, SignalPlanesObsFunction
            self.native = _native
            cfg = {'num_rbs': r, 'num_cues': c, 'num_due_pairs': p, 'device_ordinal': local,
                   'obs_fn': {'linear': LinearObsFunction, 'table': OwnLinkObsFunction, 'none': SignalPlanesObsFunction}[obs]}
            if w.get('plugin'):
                from gym_d2d_amd.path_loss import FreeSpacePathLoss
                cfg['path_loss_model'] = FreeSpacePathLoss
            if w.get('hata'):
                from gym_d2d_amd.path_loss import CostHataPathLoss
                cfg['path_loss_model'] = CostHataPathLoss
            if self.f64:
                cfg['obs_dtype'] = 'float64'
            self.reward_per_env = bool(getattr(args, 'reward_per_env', False) if reward_per_env is None else reward_per_env) and obs != 'linear'
            self.env = VecD2DEnv(cfg, num_envs=b, first_env=rank * b, cue_actions=self.cue_mode, export_actions=export,
                                 reward_per_env=self.reward_per_env, placement_trials=placement_trials)
            self.h = h = self.env.simulator.handle
            self.n_agents = self.env.num_agents
            tune_keys = {'rows': _native.TUNE_OBS_ROWS_PER_WG, 'nt': _native.TUNE_OBS_NONTEMPORAL,
                         'xcd': _native.TUNE_OBS_XCD_REMAP, 'block': _native.TUNE_OBS_BLOCK,
                         'threads': _native.TUNE_STEP_THREADS, 'epw': _native.TUNE_STEP_ENVS_PER_WG,
                         'sblock': _native.TUNE_STEP_BLOCK, 'fuse': _native.TUNE_STEP_FUSE_OBS, 'walk': _native.TUNE_STEP_WALK,
                         'prefetch': _native.TUNE_STEP_PREFETCH, 'lpt': _native.TUNE_STEP_LPT, 'variant': _native.TUNE_OBS_VARIANT}
            for kv in filter(None, tune.split(',')):
                k, v = kv.split('=')
                if k == 'bucket':
                    h.set_bucketing(bool(int(v)))
                else:
                    h.set_tuning(tune_keys[k], int(v))
            self.env.reset(seed=1234)
        g = torch.Generator(device=dev)
        g.manual_seed(1234 + rank)
        pc, pd = self.env.num_pwr_actions['cue'], self.env.num_pwr_actions['due']
        # actions of the links the AGENTS drive: [total, B, C+P], or DUE-only [total, B, P] when the CUEs follow the traffic model
        # (action_pool > 0: that many distinct action tensors, walked round-robin - long runs of a large workload)
        self.pool = pool = min(action_pool, self.total) if action_pool > 0 else self.total
        self.actions = torch.empty((pool, b, self.n_agents), dtype=torch.int32, device=dev)
        if self.n_agents == self.n and c:
            self.actions[:, :, :c] = torch.randint(0, r * pc, (pool, b, c), generator=g, device=dev, dtype=torch.int32)
        if p:
            self.actions[:, :, self.n_agents - p:] = torch.randint(0, r * pd, (pool, b, p), generator=g, device=dev, dtype=torch.int32)
        # An event pair around ONE launch adds about 3-6 us to what it measures (tools/probes/launch_floor.hip) and costs host
        # time per step: irrelevant beside the 3.7 ms obs kernel, 10-30 % of a 17-40 us step.  So: two kernels per step with
        # the expansion dominant (LinearObs, N > 128) -> per-launch events inside the timed region.  One kernel per step
        # (compact table, or small N with the expansion fused) -> the timed region runs without events, and a second pass
        # over the same steps brackets GROUPS of back-to-back launches with one event pair (on the stream the kernels run
        # on): average launch duration = group time / launches, inter-launch gaps included.
        self.events_in_timed = self.obs == 'linear' and self.n > 128
        self.raw_handle = False                          # core_mode switches the handle's obs mode under the env and sets this

    def run(self, k0, k1, gatherer=None, with_reset=False):
        """Steps k0 .. k1-1 through the PUBLIC batched API, VecD2DEnv.step(actions) -> (obs, rewards, dones, info) (the C-ABI
        handle directly only where the session's obs mode was switched under the env: core_mode; and for the CPU stub)."""
        h, env, actions = self.h, self.env, self.actions
        step = (lambda a: h.step(a.data_ptr())) if (self.stub or self.raw_handle) else env.step
        for k in range(k0, k1):
            new_episode = with_reset and k % 10 == 0
            if new_episode:                              # EPISODE_LENGTH = 10 (d2d_env.py:16): new layout per episode
                h.reset_positions(1234, k // 10)
            step(actions[k % self.pool])
            planes = getattr(self, 'planes_gather', None)
            if gatherer is not None and (new_episode or k == 0):
                # position columns only change at reset (not per step); planes plan: the library's own link rows
                gatherer.gather_positions(planes[3]() if planes else (env.link_positions() if self.obs == 'none' else env._t['table']))
            if gatherer is not None and planes:
                gatherer.launch(planes[0], sinr=planes[1], snr=planes[2])
            elif gatherer is not None:
                if self.obs == 'none':
                    gatherer.launch(env._t['reward'], sinr=env._t['sinr_db'], snr=env._t['snr_db'])
                else:
                    gatherer.launch(env._t['reward'], env._t['table'])
        if gatherer is not None:
            gatherer.wait()

    def timed(self, fence, gatherer=None, with_reset=False):
        """W untimed steps, then exactly K steps between two fences (barrier + device synchronise on both sides)."""
        torch, h = self.torch, self.h
        self.run(0, self.warmup, gatherer, with_reset)
        fence()
        if gatherer is not None:
            gatherer.reset_timing()
        h.profile_reset()
        h.profile_enable(self.events_in_timed)
        t0 = time.perf_counter()
        self.run(self.warmup, self.total, gatherer, with_reset)
        fence()
        dt = time.perf_counter() - t0
        step_ms, step_n = h.profile_read(0)
        obs_ms, obs_n = h.profile_read(1)
        step_med = h.profile_median(0) if self.events_in_timed and not self.stub else 0.0
        obs_med = h.profile_median(1) if self.events_in_timed and not self.stub else 0.0
        h.profile_enable(False)
        if not self.events_in_timed and not self.stub:
            stream = torch.cuda.current_stream(self.dev)          # VecD2DEnv runs the library's kernels on this stream
            pairs = []
            for k0 in range(self.warmup, self.total, GROUP):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                k1 = min(k0 + GROUP, self.total)
                e0.record(stream)
                for k in range(k0, k1):
                    h.step(self.actions[k % self.pool].data_ptr())
                e1.record(stream)
                pairs.append((e0, e1, k1 - k0))
            torch.cuda.synchronize(self.dev)
            step_ms = sum(e0.elapsed_time(e1) for e0, e1, _ in pairs)
            step_n = sum(c for _, _, c in pairs)
            obs_ms, obs_n = 0.0, 0
            per_group = sorted(e0.elapsed_time(e1) / c for e0, e1, c in pairs)
            step_med, obs_med = (per_group[len(per_group) // 2] if per_group else 0.0), 0.0
            if per_group:
                q = lambda f: per_group[min(len(per_group) - 1, int(f * len(per_group)))] * 1e3
                self.launch_distribution_us = {'min': q(0.0), 'p10': q(0.1), 'median': q(0.5), 'p90': q(0.9), 'max': per_group[-1] * 1e3,
                                               'groups': len(per_group), 'launches_per_group': GROUP,
                                               'what': f'average launch duration of each group of {GROUP} back-to-back launches (one HIP-event pair per '
                                                       'group: a pair around ONE launch adds 3 - 6 us to what it measures), over the timed steps'}
        return {'dt': dt, 'step_ms': step_ms, 'step_n': step_n, 'obs_ms': obs_ms, 'obs_n': obs_n, 'step_median_ms': step_med, 'obs_median_ms': obs_med}

    def roofline(self, t):
        """The dominant kernel's block: algorithmic bytes per launch / its average launch duration (HIP events); the median
        launch beside the mean (per launch where every launch has its own event pair, per group of launches otherwise)."""
        b, n = self.b, self.n
        fused = self.obs == 'linear' and not self.events_in_timed   # small N: the expansion runs inside the step launch
        if fused:
            per_launch = b * n * (CORE_BYTES + 24.0 * n)
            avg_ms, med_ms = t['step_ms'] / max(t['step_n'], 1), t.get('step_median_ms', 0.0)
            roof = {'kernel': 'step_kernel (LinearObs expansion fused)'}
        elif self.obs == 'linear' and t['obs_n']:
            # dominant kernel: obs expansion.  Algorithmic bytes per launch = B*N*(24N written (48N as float64) + 24 read of T)
            per_launch = b * n * ((48.0 if self.f64 else 24.0) * n + 24.0)
            avg_ms, med_ms = t['obs_ms'] / t['obs_n'], t.get('obs_median_ms', 0.0)
            roof = {'kernel': 'obs_expand_flat_f64_kernel' if self.f64 else 'obs_expand_flat_kernel'}
        else:
            # what the TIMED configuration moves: the decoded (rb, pwr) planes count when it writes them
            per_launch = b * n * algorithmic_bytes(n, self.obs, getattr(self, 'reward_per_env', False), self.current_export())
            avg_ms, med_ms = t['step_ms'] / max(t['step_n'], 1), t.get('step_median_ms', 0.0)
            roof = {'kernel': 'rollout_kernel / step_kernel'}
        ach = per_launch / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        roof.update({'bound': 'hbm', 'achieved': ach, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': ach / HBM_PEAK_GBS,
                     'avg_launch_ms': avg_ms, 'median_launch_ms': med_ms or None,
                     'frac_at_median_launch': (per_launch / (med_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if med_ms > 0 else None,
                     'algorithmic_bytes_per_launch': per_launch, 'traffic': None,
                     'timing': ('HIP events around every launch on the library stream, inside the timed region' if self.events_in_timed
                                else f'HIP events around groups of {GROUP} back-to-back launches (one kernel per step) on the '
                                     'library stream, in a second pass over the same steps; the timed region itself runs '
                                     'without events; median = the median group')})
        if not fused and not (self.obs == 'linear' and t['obs_n']) and self.obs == 'table' and self.current_export():
            # SURVEY.md 8(d) prices the compact-table step at 40 + 24 = 64 bytes per link; this configuration also writes the decoded
            # (rb, pwr) planes (72): the same launch on the 64-byte accounting, beside it
            roof['frac_on_64B_accounting'] = (b * n * 64.0 / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if avg_ms > 0 else None
        if getattr(self, 'launch_distribution_us', None) and not self.events_in_timed:
            roof['launch_distribution_us'] = self.launch_distribution_us
        if not self.stub:
            attach_traffic(roof, self.key, self.obs + ('_f64' if self.f64 else '') + ('_per_env_reward' if getattr(self, 'reward_per_env', False) else ''),
                           bool(self.args.envs), self.current_export())
        return roof

    def current_export(self):
        return getattr(self, 'export_now', self.export)

    def config(self, world, gather_desc=''):
        w = self.w
        return {'workload': w['name'], 'envs_per_gpu': self.b, 'links_per_env': self.n, 'obs_mode': self.obs,
                'reward_fn': 'SystemCapacity',
                'path_loss': 'FreeSpacePathLoss (plugin class)' if w.get('plugin') else ('CostHataPathLoss (suburban; per-device exponents 3.5 - 4.4)' if w.get('hata') else 'LogDistance(ple=2)'),
                'actions': 'fresh i.i.d. per step (pre-generated in HBM)',
                'api': 'C-ABI handle (d2d_step)' if (self.stub or self.raw_handle) else 'VecD2DEnv.step(actions) -> (obs, rewards, dones, info), the public batched API',
                'decoded_rb_pwr_export': 'on' if self.export else 'off (d2d_set_export_actions(0))',
                'cue_actions': self.cue_mode + (' (UplinkTrafficModel round-robin, held in the kernel\'s link records; agents supply DUE actions only)'
                                                if self.cue_mode == 'traffic' else ' (agents supply CUE and DUE actions)'),
                'parallelism': f'env-shard x{world}' + gather_desc}

    def vec_env_step_ms(self, steps=200):
        """Wall time per VecD2DEnv.step - the PUBLIC batched API (obs, rewards, dones, info out), fresh actions every step,
        nothing synchronised inside the loop - beside the bare C-ABI handle loop over the same steps."""
        torch, env, h = self.torch, self.env, self.h
        acts = [self.actions[k % self.pool] for k in range(steps)]
        out = {}
        for name, fn in (('vec_env_step', lambda a: env.step(a)), ('handle_step', lambda a: h.step(a.data_ptr()))):
            for a in acts[:10]:
                fn(a)
            torch.cuda.synchronize(self.dev)
            t0 = time.perf_counter()
            for a in acts:
                fn(a)
            host = time.perf_counter() - t0               # the loop's own (host) time: what the Python wrapper costs
            torch.cuda.synchronize(self.dev)
            out[name + '_ms'] = (time.perf_counter() - t0) / steps * 1e3
            out[name + '_host_ms'] = host / steps * 1e3
        return out

    def close(self):
        self.env.close()
        self.actions = None


def summarise(sess, t, steps, world=1):
    """A workload's entry under other_workloads / core_mode: throughput, ms per step and the roofline block."""
    return {'workload': sess.w['name'], 'obs_mode': sess.obs + (' (float64)' if sess.f64 else ''), 'cue_actions': sess.cue_mode, 'steps': steps,
            'value': sess.b * sess.n * steps * world / t['dt'], 'unit': 'agent-steps/s', 'ms_per_step': t['dt'] / steps * 1e3,
            'algorithmic_bytes_per_agent_step': algorithmic_bytes(sess.n, sess.obs, getattr(sess, 'reward_per_env', False),
                                                                  sess.current_export() and sess.obs != 'linear', sess.f64),
            'roofline': sess.roofline(t)}


def worker(args):
    import torch
    rank = int(os.environ.get('RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    local = int(os.environ.get('LOCAL_RANK', 0))
    if world != args.gpus:
        raise SystemExit(f'WORLD_SIZE={world} but --gpus {args.gpus}')
    if os.environ.get('D2D_BENCH_TEST_FAIL_RANK') == str(rank):       # test hook: a rank that dies must fail the job
        raise SystemExit(3)
    if os.environ.get('D2D_BENCH_TEST_HANG_RANK') == str(rank):       # test hook: a rank that never returns must not hold the job
        time.sleep(3600)
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    stub = args.stub_cpu
    w = dict(WORKLOADS[args.workload])
    if args.envs:
        w['envs'] = args.envs

    # the CPU baseline runs first: this process has not initialised HIP yet, so its fork()ed pool is safe
    cpu = None
    if world == 1 and rank == 0 and not args.no_cpu_baseline and not stub:
        cpu = cpu_baseline(w)

    # stdout carries exactly ONE line (the JSON): RCCL prints a version banner to fd 1 when its first communicator is
    # created, so everything until the result is ready goes to stderr instead
    sys.stdout.flush()
    saved_stdout = os.dup(1)
    os.dup2(2, 1)
    if stub:
        dev = torch.device('cpu')
    else:
        if args.share_gpu:
            local = 0
        if local >= torch.cuda.device_count():
            raise SystemExit(f'rank {rank}: needs GPU {local} but this box exposes {torch.cuda.device_count()}')
        torch.cuda.set_device(local)
        dev = torch.device('cuda', local)
    use_dist = world > 1 or args.force_dist
    if use_dist:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if 'MASTER_PORT' not in os.environ:
            if world > 1:
                raise SystemExit('MASTER_PORT is not set: launch the ranks through `python bench.py --gpus N` or torch.distributed.run')
            os.environ['MASTER_PORT'] = str(_free_port())          # one rank (--force-dist): any free port will do
        if stub or args.share_gpu:
            dist.init_process_group('gloo', rank=rank, world_size=world)
        else:
            dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)

    # the path every existing gym-d2d user calls: one env, dicts in / out (d2d_env.py:62-71).  Reported beside the batch
    # number, never as `value`; timed before the batch env exists so its 26 GB of buffers are not in the picture.
    c, p, r = w['cues'], w['dues'], w['rbs']
    single_ms = None
    if rank == 0 and world == 1 and not args.no_single_env_latency and not stub:
        single_ms = {'25 CUE + 25 DUE pairs, 25 RB (reference default env)': single_env_latency(25, 25, 25, local)}
        if (c, p, r) != (25, 25, 25):
            single_ms[f'{c} CUE + {p} DUE pairs, {r} RB'] = single_env_latency(c, p, r, local)
            single_ms[f'{c} CUE + {p} DUE pairs, {r} RB, custom Python PathLoss (table route)'] = custom_path_loss_cost(c, p, r, local)

    def fence():
        if not stub:
            torch.cuda.synchronize(dev)
        if use_dist:
            dist.barrier()
        if not stub:
            torch.cuda.synchronize(dev)

    sess = Session(torch, args, args.workload, args.obs, dev, rank, local, args.steps, args.warmup, envs=args.envs,
                   cue_mode=args.cue_actions, tune=args.tune, stub=stub, export=not args.no_export, f64=args.obs_dtype == 'float64',
                   action_pool=128 if args.steps + args.warmup > 128 else 0)      # long runs: 128 action tensors (1 GB at 4096 x 512) round-robin
    b, n = sess.b, sess.n

    def make_gatherer(mode, signal_every=1):
        from gym_d2d_amd.distributed import StepGatherer
        native = args.gather_backend == 'native' and not stub and not args.share_gpu
        return StepGatherer(b, n, dev, mode=mode, signal_every=signal_every, timing=not stub,
                            reward_every=args.reward_every if mode == 'rewards' else 1,
                            backend='native' if native else 'torch', handle=sess.h if native else None)

    if args.obs == 'none' and args.gather == 'table':
        args.gather = 'planes'                 # no table exists in the obs-less mode: the (sinr, snr) planes travel instead (same bytes)
    gatherer = make_gatherer(args.gather, args.signal_every) if use_dist and not args.no_gather else None
    total_steps = args.steps + args.warmup                    # launches the gatherer sees in the timed run (warm-up included)
    t = sess.timed(fence, gatherer, args.with_reset)
    flags = sess.env.status_flags()
    gather_ms = gatherer.gather_ms() if gatherer is not None else None
    # evidence that the collective saw every rank: a count all-reduce, and the last step's rewards summed two ways
    # (all-reduce of the local sums vs the sum of what the all-gather delivered)
    dist_info = {}
    ring = gatherer is not None and gatherer.reward_every > 1
    if use_dist:
        ones = torch.ones(1, device=dev, dtype=torch.float64)
        dist.all_reduce(ones)
        rew = sess.env._t['reward']
        local_sum = (rew if rew.dim() == 1 else rew[:, 0]).double().sum().reshape(1)
        dist.all_reduce(local_sum)
        dist_info = {'rccl_ranks': dist.get_world_size(), 'backend': dist.get_backend(),
                     'allreduce_rank_count': float(ones.item()), 'allreduce_reward_checksum': float(local_sum.item())}
        if gatherer is not None:
            all_reward, all_signal = gatherer.wait()
            if ring:                                          # [K, B_global] block: its last row is the last step only if the run ended on a block boundary
                idx = total_steps - 1 - gatherer.reward_step
                all_reward = all_reward[idx] if gatherer.reward_step >= 0 and 0 <= idx < gatherer.reward_every else None
            gsum = float(all_reward.double().sum().item()) if all_reward is not None else float('nan')
            dist_info['allgather_reward_checksum'] = gsum
            dist_info['allgather_envs'] = int(all_reward.numel()) if all_reward is not None else 0
            dist_info['checksums_agree'] = bool(abs(gsum - dist_info['allreduce_reward_checksum'])
                                                <= 1e-6 * max(1.0, abs(gsum))) if all_reward is not None else None
            if all_reward is None:
                dist_info['checksums_note'] = ('the last step is not the end of a gathered reward block (steps + warmup is not a multiple of '
                                               '--reward-every): nothing to compare')
    # what the gather costs the step loop: the same steps again without it
    t_nogather = sess.timed(fence, None, args.with_reset) if gatherer is not None else None

    # report the 'core' mode beside the with-obs mode.  Same envs, same actions, obs_fn swapped for the
    # compact table (what a learner that builds its own features consumes): the step is then the step kernel alone.
    # The decoded (rb, pwr) planes behind info['rb'] / info['tx_pwr_dbm'] are a rollout's own actions again: the entry is
    # measured without them (d2d_set_export_actions(0), 64 algorithmic bytes per link) and, beside it, with them.
    core = None
    if not stub and sess.obs == 'linear' and not sess.f64 and n > 128 and (world == 1 or not args.no_gather):
        core = core_mode(torch, sess, dev, args, fence, use_dist, world, make_gatherer)

    dt = t['dt']
    if use_dist:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
        # every rank's own clock and kernels, rank-major
        mine = torch.tensor([t['dt'] / args.steps * 1e3, t['step_ms'] / max(t['step_n'], 1), (t['obs_ms'] / t['obs_n']) if t['obs_n'] else 0.0,
                             gather_ms or 0.0], device=dev, dtype=torch.float64)
        every = torch.empty(world * 4, device=dev, dtype=torch.float64)
        dist.all_gather_into_tensor(every, mine)
        every = every.view(world, 4)
        dist_info['per_rank'] = [{'rank': k, 'ms_per_step': float(row[0]), 'step_kernel_ms': float(row[1]),
                                  'obs_expand_kernel_ms': float(row[2]) or None, 'gather_ms_per_step': float(row[3]) or None}
                                 for k, row in enumerate(every.cpu())]
        if gatherer is not None:
            tn = torch.tensor([t_nogather['dt']], device=dev, dtype=torch.float64)
            dist.all_reduce(tn, op=dist.ReduceOp.MAX)
            dist_info['gather'] = {
                'mode': args.gather, 'signal_every': args.signal_every, 'reward_every': gatherer.reward_every, 'backend': gatherer.backend,
                'bytes_per_gpu_per_step': gatherer.bytes_per_signal_launch if args.signal_every == 1 else
                gatherer.bytes_per_launch + (gatherer.bytes_per_signal_launch - gatherer.bytes_per_launch) / args.signal_every,
                'gather_ms_per_step': gather_ms,                                   # side-stream events, rank 0
                'ms_per_step_without_gather': float(tn.item()) / args.steps * 1e3,  # max over ranks
                'gather_exposed_ms': (dt - float(tn.item())) / args.steps * 1e3,    # what the step loop pays for it
                'timing': 'HIP events on the gather side stream around (staging copies + all-gathers) of every launch; exposed = '
                          'ms_per_step with the gather minus the same steps without it'}

    extras = {}
    if rank == 0 and world == 1 and not stub and not args.no_extras and args.workload == 'stress' and args.obs == 'linear' \
            and not args.envs and not args.tune and not sess.f64:
        sess.close()
        extras = n1_extras(torch, args, dev, local, fence)

    if rank == 0:
        agent_steps = b * n * args.steps * world
        value = agent_steps / dt
        roof = sess.roofline(t)
        if extras.get('box_write_ceiling'):
            roof['box_ceiling_GBs'] = extras['box_write_ceiling']['GBps']
            roof['frac_of_box_ceiling'] = roof['achieved'] / roof['box_ceiling_GBs'] if roof['kernel'].startswith('obs_expand') else None
            roof['box_ceiling_note'] = ('the best store-only kernel this library could build on this box (box_write_ceiling: plain fills in 32 geometries, '
                                        'hipMemsetAsync, and fills with the obs kernel\'s load + barrier structure and scope-bit stores)')
        carried = {'table': ', (sinr, snr) columns of the obs table', 'planes': ', the (sinr, snr) result planes', 'rewards': ''}[args.gather]
        if carried and args.signal_every > 1:
            carried += f' every {args.signal_every} steps'
        cfg = sess.config(world, f' + per-step all-gather ({args.gather}: rewards{carried}; position columns once per episode)' if gatherer else '')
        cfg['positions'] = 'redrawn on the device every 10 steps' if args.with_reset else 'fixed over the run'
        out = {
            'metric': 'env agent-steps/sec (batch x agents)', 'value': value, 'unit': 'agent-steps/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': dt / args.steps * 1e3,
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'obs_dtype': 'float64' if sess.f64 else 'float32',
            'config': cfg,
            'roofline': roof,
            'kernels': {'step_kernel_ms': t['step_ms'] / max(t['step_n'], 1), 'obs_expand_kernel_ms': (t['obs_ms'] / t['obs_n']) if t['obs_n'] else None},
            'algorithmic_bytes_per_agent_step': algorithmic_bytes(n, sess.obs, getattr(sess, 'reward_per_env', False),
                                                                  sess.export and sess.obs != 'linear', sess.f64),
            'status_flags': flags,
            'target_agent_steps_per_s': 1e7,
        }
        if world > 1 or args.force_dist:
            out['value_per_gpu'] = value / world
        out.update(dist_info)
        if stub:
            out['stub'] = 'CPU plumbing test - not a measurement'
        if args.share_gpu:
            out['shared_gpu'] = f'{world} ranks on ONE GPU over gloo: multi-rank correctness test - not a measurement'
        if single_ms is not None:
            out['single_env_step_ms'] = single_ms
        if core is not None:
            out['core_mode'] = core
        for k in ('other_workloads', 'vec_env_step_ms', 'box_write_ceiling'):
            if k in extras:
                out[k] = extras[k]
        # the secondary figures as FLAT scalars (a parser that keeps only top-level scalars still carries them)
        def _frac(d, *path):
            for k in path:
                d = d.get(k) if isinstance(d, dict) else None
            return d.get('frac') if isinstance(d, dict) else None
        ow = extras.get('other_workloads', {})
        flat = {'frac_config2': _frac(ow, 'default', 'roofline'), 'us_per_step_config2': (ow.get('default') or {}).get('ms_per_step'),
                'frac_config2_with_placement_trials': (ow.get('default') or {}).get('with_placement_trials_24', {}).get('frac'),
                'frac_config4': _frac(ow, 'plugin', 'roofline'),
                'frac_config4_on_64B': ((ow.get('plugin') or {}).get('roofline') or {}).get('frac_on_64B_accounting'),
                'frac_power_law_obsless': _frac(ow, 'power_law_obsless', 'roofline'),
                'us_config2_launch_min_median_p90': [((ow.get('default') or {}).get('roofline', {}).get('launch_distribution_us') or {}).get(k) for k in ('min', 'median', 'p90')]
                if ((ow.get('default') or {}).get('roofline', {}).get('launch_distribution_us')) else None,
                'frac_obs_float64': _frac(ow, 'stress_float64_obs', 'roofline'),
                'frac_table_no_export': _frac(core or {}, 'roofline'), 'frac_table_with_export': _frac(core or {}, 'with_decoded_rb_pwr_export', 'roofline'),
                'frac_obsless': _frac(core or {}, 'planes_only', 'roofline'),
                'us_step_kernel_obsless': ((core or {}).get('planes_only', {}).get('roofline', {}) or {}).get('avg_launch_ms'),
                'median_launch_ms': roof.get('median_launch_ms'), 'frac_at_median_launch': roof.get('frac_at_median_launch')}
        if flat['us_per_step_config2'] is not None:
            flat['us_per_step_config2'] *= 1e3
        if flat['us_step_kernel_obsless'] is not None:
            flat['us_step_kernel_obsless'] *= 1e3
        if single_ms is not None:
            flat['single_env_step_ms_default'] = single_ms.get('25 CUE + 25 DUE pairs, 25 RB (reference default env)')
        out.update({k: v for k, v in flat.items() if v is not None})
        if cpu is not None:
            out['cpu_baseline'] = cpu
        elif world > 1:
            out['cpu_baseline'] = 'N=1 line only (the CPU port is timed on rank 0 of the one-GPU run)'
        sys.stdout.flush()
        os.dup2(saved_stdout, 1)                 # the real stdout is back for the one line that belongs there
        print(json.dumps(out), flush=True)
        os.dup2(2, 1)                            # teardown chatter (RCCL / ROCm) goes to stderr again
    if sess.actions is not None:
        sess.close()
    if use_dist:
        dist.destroy_process_group()


def core_mode(torch, sess, dev, args, fence, use_dist, world, make_gatherer):
    """The headline session's envs and actions stepped with the compact-table obs (no LinearObs expansion); with N > 1 the
    per-step gather is the rewards-only plan (16 KB per GPU: a 16.8 MB ring all-gather would outlast the 30 us step ten times)."""
    _native, h, b, n = sess.native, sess.h, sess.b, sess.n
    h.set_obs_mode(_native.OBS_TABLE)
    saved_obs, saved_events = sess.obs, sess.events_in_timed
    sess.obs, sess.events_in_timed, sess.raw_handle = 'table', False, True
    out = {}
    try:
        for export in (False, True):
            h.set_export_actions(export)
            sess.export_now = export
            gatherer = make_gatherer('rewards') if use_dist else None
            t = sess.timed(fence, gatherer)
            dt = t['dt']
            if use_dist:
                import torch.distributed as dist
                tt = torch.tensor([dt], device=dev, dtype=torch.float64)
                dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                dt = float(tt.item())
            entry = {'obs_mode': 'table (compact [B, N, 6] obs, no LinearObs expansion)', 'value': b * n * args.steps * world / dt,
                     'unit': 'agent-steps/s', 'ms_per_step': dt / args.steps * 1e3, 'roofline': sess.roofline(t)}
            if gatherer is not None:
                entry['gather'] = {'mode': 'rewards', 'reward_every': gatherer.reward_every, 'bytes_per_gpu_per_step': gatherer.bytes_per_launch,
                                   'gather_ms_per_step': gatherer.gather_ms()}
            if not export:
                entry['decoded_rb_pwr_export'] = ('off (d2d_set_export_actions(0)): info[rb] / info[tx_pwr_dbm] are the rollout\'s own '
                                                  'actions; 64 algorithmic bytes per link and step')
                out = entry
            else:
                out['with_decoded_rb_pwr_export'] = {k: entry[k] for k in ('value', 'ms_per_step', 'roofline')}
        # The learner configuration of SURVEY.md 8(e): no observation array at all (D2D_OBS_NONE - the (sinr, snr) planes ARE the
        # per-step observation, positions once per reset from D2D_BUF_LINK_POS), the env's reward once per env
        # (D2D_REWARD_PER_ENV), no decoded (rb, pwr) planes.  Its own algorithmic bytes: action 4 + positions 16 + four result
        # planes 16 + 4 / N of reward = 36 per link and step.  With N > 1 the gather is StepGatherer(mode='planes').
        h.set_obs_mode(_native.OBS_NONE)
        h.set_export_actions(False)
        h.set_reward_layout(_native.REWARD_PER_ENV)
        sess.obs, sess.export_now, sess.reward_per_env = 'none', False, True
        renv = torch.empty(b, dtype=torch.float32, device=dev)
        h.bind_buffer(_native.BUF_REWARD_ENV, renv.data_ptr(), b * 4)
        try:
            gatherer = None
            if use_dist:
                from gym_d2d_amd.distributed import StepGatherer
                gatherer = StepGatherer(b, n, dev, mode='planes', signal_every=max(args.signal_every, 1), timing=True)
                sess.planes_gather = (renv, sess.env._t['sinr_db'], sess.env._t['snr_db'], sess.env.link_positions)
            t = sess.timed(fence, gatherer)
            dt = t['dt']
            if use_dist:
                import torch.distributed as dist
                tt = torch.tensor([dt], device=dev, dtype=torch.float64)
                dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                dt = float(tt.item())
            out['planes_only'] = {
                'obs_mode': 'none (D2D_OBS_NONE: the sinr_db / snr_db planes are the per-step observation; positions once per reset '
                            'from D2D_BUF_LINK_POS), reward once per env (D2D_REWARD_PER_ENV), no decoded (rb, pwr) planes',
                'value': b * n * args.steps * world / dt, 'unit': 'agent-steps/s', 'ms_per_step': dt / args.steps * 1e3,
                'algorithmic_bytes_per_agent_step': algorithmic_bytes(n, 'none', True), 'roofline': sess.roofline(t)}
            if gatherer is not None:
                out['planes_only']['gather'] = {'mode': 'planes', 'signal_every': gatherer.signal_every,
                                                'bytes_per_gpu_per_step': gatherer.bytes_per_signal_launch,
                                                'gather_ms_per_step': gatherer.gather_ms()}
        finally:
            sess.planes_gather = None
            sess.reward_per_env = False
            h.set_reward_layout(_native.REWARD_PER_AGENT)
            h.bind_buffer(_native.BUF_REWARD_ENV, 0, 0)
    finally:
        h.set_export_actions(True)
        sess.export_now = sess.export
        h.set_obs_mode(_native.OBS_LINEAR)
        sess.obs, sess.events_in_timed, sess.raw_handle = saved_obs, saved_events, False
    return out


def n1_extras(torch, args, dev, local, fence):
    """Beside the headline, in the same driver-run line (never as `value`): BASELINE configs 2 and 4 and an episode loop with
    the device-side reset, each with its own roofline block; the public VecD2DEnv.step against the bare handle loop; the
    box's write ceiling."""
    out = {'other_workloads': {}, 'vec_env_step_ms': {}}
    sys.path.insert(0, str(ROOT / 'tools'))
    import write_probe                                    # libd2d_probe.so: measurement equipment, not in the product library
    # BASELINE.json configs[1]: 1024 x 50, traffic-model CUEs, LinearObs fused into the step launch.  2000 + 2000 steps (60 ms):
    # behind an idle stretch (closing one session and building the next is host work) the chip needs about 1000 of these
    # 14 us steps (15 ms) to come back to its busy clocks and runs 5 - 10 % slower until then (tools/probes/clock_state.py,
    # profiles/r4_clock_state_default.jsonl) - a 220-step run measured the ramp, not the kernel
    s = Session(torch, args, 'default', 'linear', dev, 0, local, 2000, 2000, action_pool=256)
    out['other_workloads']['default'] = summarise(s, s.timed(fence), 2000)          # a plain VecD2DEnv(cfg, 1024): no placement trials
    # which lottery ticket this box / allocation drew (the 0.53 <-> 0.61 spread is where the 61 MB obs block sits
    # physically - reproducible per allocation, predicted by nothing observable, profiles/NEGATIVE_RESULTS.md "Config 2")
    med = (out['other_workloads']['default']['roofline'].get('launch_distribution_us') or {}).get('median')
    if med:
        out['other_workloads']['default']['roofline']['placement_class'] = (
            ('fast' if med <= 13.5 else ('middle' if med < 14.5 else 'slow')) + f' (median group {med:.2f} us; classes seen on this pool: '
            'about 13.2 / 13.9 / 15.1 us, profiles/r4_obs_block_placement_candidates.jsonl)')
    try:                                                  # what a pure fill of about the same size reaches on this box (61 MB of obs per step)
        small, _ = write_probe.write_variants(64 << 20, 20, local)
        out['other_workloads']['default']['roofline']['box_ceiling_GBs_64MiB_bursts'] = small
    except Exception as exc:                              # pragma: no cover
        out['other_workloads']['default']['roofline']['box_ceiling_error'] = repr(exc)
    out['vec_env_step_ms']['default (config 2), LinearObs'] = s.vec_env_step_ms()
    s.close()
    # ... and beside it, never as the entry's own figure: the same env built with the OPT-IN placement trials (where a 61 MB obs
    # block lands decides its speed class - 13.2 / 13.9 / 15.1 us per step, profiles/r4_obs_block_placement_candidates.jsonl;
    # VecD2DEnv(placement_trials=24) times candidate blocks at its first reset and keeps the fastest)
    try:
        s = Session(torch, args, 'default', 'linear', dev, 0, local, 2000, 2000, action_pool=256, placement_trials=24)
        try:
            e = summarise(s, s.timed(fence), 2000)
            out['other_workloads']['default']['with_placement_trials_24'] = {
                'value': e['value'], 'ms_per_step': e['ms_per_step'], 'frac': e['roofline']['frac'], 'avg_launch_ms': e['roofline']['avg_launch_ms'],
                'trials': getattr(s.env, 'placement', None)}
        finally:
            s.close()
    except Exception as exc:                              # pragma: no cover
        out['other_workloads']['default']['with_placement_trials_24'] = {'error': repr(exc)}
    # BASELINE.json configs[3]: FreeSpacePathLoss + a custom ObsFunction through the plugin ABI, 4096 x 512
    s = Session(torch, args, 'plugin', 'table', dev, 0, local, 1000, 1000, action_pool=64)      # 58 ms: as above
    out['other_workloads']['plugin'] = summarise(s, s.timed(fence), 1000)
    out['vec_env_step_ms']['stress sizes, compact obs (OwnLinkObsFunction)'] = s.vec_env_step_ms()
    s.close()
    # an exponent other than 2 (COST-Hata: rows a5, a5''): the power-law rollout kernel in the obs-less learner mode
    try:
        s = Session(torch, args, 'hata', 'none', dev, 0, local, 1000, 1000, action_pool=64, export=False, reward_per_env=True)
        try:
            out['other_workloads']['power_law_obsless'] = summarise(s, s.timed(fence), 1000)
        finally:
            s.close()
    except Exception as exc:                              # pragma: no cover
        out['other_workloads']['power_law_obsless'] = {'error': repr(exc)}
    # the stress workload as EPISODES: positions redrawn on the device every 10 steps (d2d_env.py:16,45-52), 3 episodes
    s = Session(torch, args, 'stress', 'linear', dev, 0, local, 30, 10)
    e = summarise(s, s.timed(fence, None, True), 30)
    e['positions'] = 'redrawn on the device every 10 steps (reset_kernel inside the timed region), 3 episodes'
    out['other_workloads']['stress_with_reset'] = e
    s.close()
    # the reference's own observation dtype (obs_fn.py:47,51: float64 arrays), written as float64 by the expansion kernel itself:
    # 48 N bytes per agent-step, a 51.5 GB block
    try:
        s = Session(torch, args, 'stress', 'linear', dev, 0, local, 30, 5, f64=True)
        try:
            out['other_workloads']['stress_float64_obs'] = summarise(s, s.timed(fence), 30)
        finally:
            s.close()
    except Exception as exc:                              # pragma: no cover
        out['other_workloads']['stress_float64_obs'] = {'error': repr(exc)}
    # the write ceiling of THIS box: pure fill kernels in a family of store geometries that contains the obs kernel's own
    try:
        best, rates = write_probe.write_variants(8 << 30, 5, local)
        blocks, rows = (768, 1024, 512, 256), (2, 4, 8, 32)
        k = max(range(len(rates)), key=lambda v: rates[v])
        name = 'hipMemsetAsync' if k == 32 else f'{blocks[k & 3]} threads x {rows[(k >> 2) & 3]} rows per workgroup, ' + ('plain' if k & 16 else 'nontemporal') + ' stores'
        # ... and the forms round 4 found faster than every plain fill: the obs kernel's own TIMING structure (every workgroup
        # stages a table row through LDS behind a barrier before it stores) and the gfx942+ scope bits on the stores
        staged = {}
        for label, variant in (('1024 threads x 2 rows, LDS stage + barrier, sc1 nt stores', 1 + 32 + 512),
                               ('1024 threads x 2 rows, LDS stage + barrier, sc0 sc1 nt stores', 1 + 32 + 384),
                               ('1024 threads x 2 rows, LDS stage + barrier, nt stores', 1 + 32),
                               ('768 threads x 2 rows (the obs kernel geometry), LDS stage + barrier, sc1 nt stores', 0 + 32 + 512),
                               ('768 threads x 2 rows (the obs kernel geometry), LDS stage + barrier, nt stores', 0 + 32)):
            staged[label] = max(write_probe.write_staged(8 << 30, variant, 0, iters=5, device=local) for _ in range(2))
        kb = max(staged, key=staged.get)
        if staged[kb] > best:
            best, name = staged[kb], kb
        out['box_write_ceiling'] = {'GBps': best, 'best_variant': name, 'obs_kernel_geometry_GBps': rates[0], 'hipMemsetAsync_GBps': rates[32],
                                    'best_plain_fill_GBps': max(rates[:32]), 'staged_forms_GBps': staged,
                                    'what': 'libd2d_probe.so (include/d2d_hip_diag.h), d2d_probe_write_variants: 8 GiB written 5 times by each of 32 pure fill kernels (block 768 / 1024 / 512 / 256 '
                                            'threads x 2 / 4 / 8 / 32 rows per workgroup x nontemporal / plain 16-byte stores, XCD-grouped dispatch '
                                            'order; the first is the obs kernel\'s own geometry), by hipMemsetAsync, and by the staged forms of '
                                            'd2d_probe_write_staged (a table row through LDS behind a barrier before the stores; sc1 / sc0 sc1 scope bits '
                                            'with nt); GBps = the best of them'}
    except Exception as exc:                              # pragma: no cover - a probe failure must not lose the line
        out['box_write_ceiling'] = {'error': repr(exc)}
    return out


def attach_traffic(roof, workload, obs, custom_envs, export=True):
    """HBM bytes per launch come from rocprofv3 PMC passes, which cannot be collected from inside this process: the
    figure is QUOTED from the newest committed summary for this kernel and workload, and only if that summary was
    made from the SAME kernel sources that are running now (digest of csrc/ recorded by tools/summarize_profiles.py);
    otherwise it stays null and the mismatch is reported."""
    if custom_envs:
        return
    from gym_d2d_amd.build import source_digest
    digest = source_digest()
    for path in sorted((ROOT / 'profiles').glob('r*_pmc_*.json'), reverse=True):
        try:
            rec = json.loads(path.read_text())
        except Exception:
            continue
        if rec.get('workload_key') != f'{workload}/{obs}':
            continue
        if obs != 'linear' and ('--no-export' in rec.get('command', '')) == bool(export):
            continue            # collected with / without the decoded (rb, pwr) planes: 8 bytes per link apart
        for kname, d in rec.get('kernels', {}).items():
            tag = roof['kernel'].split(' ')[0]
            tags = ('obs_expand',) if tag.startswith('obs_expand') else (('rollout_kernel', 'step_kernel') if tag == 'rollout_kernel' else (tag,))
            if any(t in kname for t in tags) and 'hbm_bytes_per_launch' in d:
                if rec.get('source_digest') == digest:
                    roof['traffic'] = d['hbm_bytes_per_launch']
                    roof['traffic_source'] = (f'QUOTED, not measured in this run: profiles/{path.name} (WRITE_SIZE + 2*FETCH_SIZE, '
                                              f'separate --pmc passes; kernel sources {digest[:12]} = the ones running)')
                else:
                    roof['traffic_stale'] = (f'profiles/{path.name} was collected for kernel sources '
                                             f'{str(rec.get("source_digest"))[:12]}, running {digest[:12]}')
                return


def single_env_latency(c, p, r, ordinal, steps=200):
    """ms per D2DEnv.step of the drop-in single env (host dicts in / out, PCIe and Python included)."""
    from gym_d2d_amd.envs import D2DEnv
    env = D2DEnv({'num_rbs': r, 'num_cues': c, 'num_due_pairs': p, 'device_ordinal': ordinal})
    obs = env.reset()
    rng = np.random.default_rng(0)
    acts = [{k: int(rng.integers(0, env.action_space['due' if k.startswith('due') else 'cue'].n)) for k in obs}
            for _ in range(8)]
    for k in range(5):
        env.step(acts[k % 8])
    import gc
    gc.collect(); gc.disable()
    try:
        t0 = time.perf_counter()
        for k in range(steps):
            env.step(acts[k % 8])
        ms = (time.perf_counter() - t0) / steps * 1e3
    finally:
        gc.enable()
    env.close()
    return ms


def custom_path_loss_cost(c, p, r, ordinal):
    """A user PathLoss that only defines __call__ (examples/custom_path_loss.py:8-16) goes through the host-evaluated table:
    what reset() costs then (the (link transmitter) x (link receiver) pairs evaluated in Python, path_loss.py table_db) and
    what a step costs afterwards."""
    import math
    from gym_d2d_amd.envs import D2DEnv
    from gym_d2d_amd.path_loss import PathLoss

    class FooPathLoss(PathLoss):
        def __call__(self, tx, rx):
            d = tx.position.distance(rx.position)
            return 20 * math.log10(d) - tx.tx_antenna_gain_dBi - rx.rx_antenna_gain_dBi
    env = D2DEnv({'num_rbs': r, 'num_cues': c, 'num_due_pairs': p, 'device_ordinal': ordinal, 'path_loss_model': FooPathLoss})
    obs = env.reset()
    t0 = time.perf_counter()
    obs = env.reset()
    reset_ms = (time.perf_counter() - t0) * 1e3
    rng = np.random.default_rng(0)
    act = {k: int(rng.integers(0, env.action_space['due' if k.startswith('due') else 'cue'].n)) for k in obs}
    env.step(act)
    t0 = time.perf_counter()
    for _ in range(10):
        env.step(act)
    step_ms = (time.perf_counter() - t0) / 10 * 1e3
    calls = len(set(env.simulator.link_tx.tolist())) * len(set(env.simulator.link_rx.tolist()))
    env.close()
    return {'reset_ms': reset_ms, 'step_ms': step_ms, 'path_loss_calls_per_reset': calls,
            'device_pairs_an_all_pairs_table_would_evaluate': (1 + c + 2 * p) ** 2}


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    args = parse_args(argv)
    if args.gpus > 1 and 'RANK' not in os.environ:
        sys.exit(launch_ranks(args.gpus, argv, args.rank_timeout))
    worker(args)


if __name__ == '__main__':
    main()
