#!/usr/bin/env python3
"""Benchmark of the hot path: D2DEnv.step for a batch of environments on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload stress|default|plugin] [--obs linear|table|none]

A "step" is one pass of the fused path over the whole batch: action decode -> SINR / SNR / rate / capacity -> reward
-> observation table -> LinearObs expansion (reference semantics: obs materialised as [B, N, 6N] in HBM), with fresh
i.i.d. actions every step (pre-generated in HBM; positions fixed for the run - the reference's step never moves
devices, simulator.py:61-75).  Metric: agent-steps/s = B * N * steps / wall seconds (whole job, all GPUs).

N > 1: one process per GPU (torch.distributed / RCCL), env axis sharded 4096 per GPU (weak scaling), one all-gather
per step of rewards + the per-step columns of the compact obs table, overlapped on a side stream.  `--gpus N` works
both ways: under `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N` (RANK is already set: this
process IS a rank) and as a plain `python bench.py --gpus N` - then this process is only a launcher: it starts the N
rank processes BEFORE anything touches HIP (it never imports torch, never re-execs), waits, and exits non-zero if any
rank failed.

Prints ONE JSON line on rank 0 with `roofline` (dominant kernel, HIP-event timed on the library's stream),
`cpu_baseline` (the NumPy fp64 oracle timed on this box's host cores on a bounded sample; rank 0, N = 1 only; one
thread, with the all-core figure beside it) and, for N > 1, `rccl_ranks` + all-reduce / all-gather checksums.  At N = 1
the line also carries, beside `value` and never as it: `core_mode` (the same envs stepped with the compact-table obs:
throughput + the step kernel's roofline block) and `single_env_step_ms` (the drop-in single-env
D2DEnv.step, host dicts in / out, at the reference's default shape and at the workload's).
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


def launch_ranks(n, argv):
    """Start one rank process per GPU and wait for them.  This process never touches HIP (torch is not even imported
    here) and never execs: the ranks are ordinary children.  Rank 0 inherits stdout (the JSON line); the other ranks'
    stdout goes to stderr.  Returns the exit code (first failing rank's, else 0)."""
    port = os.environ.get('MASTER_PORT') or str(_free_port())
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
    ap.add_argument('--envs', type=int, default=0, help='override envs per GPU')
    ap.add_argument('--cue-actions', default='', choices=['', 'agent', 'traffic'],
                    help="who drives the CUE links (default: the workload's own choice)")
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-gather', action='store_true', help='N > 1: skip the per-step all-gather')
    ap.add_argument('--with-reset', action='store_true', help='redraw all device positions every 10 steps (device-side reset)')
    ap.add_argument('--force-dist', action='store_true', help='init RCCL and run the gather path even with one rank (test hook)')
    ap.add_argument('--no-single-env-latency', action='store_true',
                    help='N = 1 also times the drop-in single-env D2DEnv.step (host dicts in / out); this skips it')
    ap.add_argument('--tune', default='', help='comma list key=value: rows,nt,xcd,bucket,block,threads,epw,sblock,fuse')
    ap.add_argument('--stub-cpu', action='store_true',
                    help='TEST HOOK: no GPU, gloo backend, synthetic per-rank results - exercises only the launcher / gather plumbing')
    ap.add_argument('--share-gpu', action='store_true',
                    help='TEST HOOK: every rank uses cuda:0 and gloo carries the collectives (RCCL refuses two ranks on one '
                         'GPU) - the real worker with N > 1 on a one-GPU box; never a measurement')
    return ap.parse_args(argv)


def worker(args):
    import torch
    rank = int(os.environ.get('RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    local = int(os.environ.get('LOCAL_RANK', 0))
    if world != args.gpus:
        raise SystemExit(f'WORLD_SIZE={world} but --gpus {args.gpus}')
    if os.environ.get('D2D_BENCH_TEST_FAIL_RANK') == str(rank):       # test hook: a rank that dies must fail the job
        raise SystemExit(3)
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    w = dict(WORKLOADS[args.workload])
    if args.envs:
        w['envs'] = args.envs
    b, c, p, r = w['envs'], w['cues'], w['dues'], w['rbs']
    n = c + p
    stub = args.stub_cpu

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
        os.environ.setdefault('MASTER_PORT', '29511')
        if stub or args.share_gpu:
            dist.init_process_group('gloo', rank=rank, world_size=world)
        else:
            dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)

    # the path every existing gym-d2d user calls: one env, dicts in / out (d2d_env.py:62-71).  Reported beside the batch
    # number, never as `value`; timed before the batch env exists so its 26 GB of buffers are not in the picture.
    single_ms = None
    if rank == 0 and world == 1 and not args.no_single_env_latency and not stub:
        single_ms = {'25 CUE + 25 DUE pairs, 25 RB (reference default env)': single_env_latency(25, 25, 25, local)}
        if (c, p, r) != (25, 25, 25):
            single_ms[f'{c} CUE + {p} DUE pairs, {r} RB'] = single_env_latency(c, p, r, local)

    if w.get('plugin'):
        args.obs = 'table'
    cue_mode = args.cue_actions or ('traffic' if w.get('traffic') else 'agent')
    if stub:
        env = _StubEnv(torch, b, n, rank * b)
        h = env.simulator.handle
        n_agents = n
    else:
        from gym_d2d_amd import _native
        from gym_d2d_amd.envs import VecD2DEnv
        from gym_d2d_amd.envs.obs_fn import LinearObsFunction, OwnLinkObsFunction
        cfg = {'num_rbs': r, 'num_cues': c, 'num_due_pairs': p, 'device_ordinal': local,
               'obs_fn': LinearObsFunction if args.obs == 'linear' else OwnLinkObsFunction}
        if w.get('plugin'):
            from gym_d2d_amd.path_loss import FreeSpacePathLoss
            cfg['path_loss_model'] = FreeSpacePathLoss
        env = VecD2DEnv(cfg, num_envs=b, first_env=rank * b, cue_actions=cue_mode)
        h = env.simulator.handle
        n_agents = env.num_agents
        if args.obs == 'none':
            h.set_obs_mode(_native.OBS_NONE)
        tune_keys = {'rows': _native.TUNE_OBS_ROWS_PER_WG, 'nt': _native.TUNE_OBS_NONTEMPORAL,
                     'xcd': _native.TUNE_OBS_XCD_REMAP, 'block': _native.TUNE_OBS_BLOCK,
                     'threads': _native.TUNE_STEP_THREADS, 'epw': _native.TUNE_STEP_ENVS_PER_WG,
                     'sblock': _native.TUNE_STEP_BLOCK, 'fuse': _native.TUNE_STEP_FUSE_OBS}
        for kv in filter(None, args.tune.split(',')):
            k, v = kv.split('=')
            if k == 'bucket':
                h.set_bucketing(bool(int(v)))
            else:
                h.set_tuning(tune_keys[k], int(v))
        env.reset(seed=1234)

    total = args.steps + args.warmup
    g = torch.Generator(device=dev)
    g.manual_seed(1234 + rank)
    pc, pd = env.num_pwr_actions['cue'], env.num_pwr_actions['due']
    # actions of the links the AGENTS drive: [total, B, C+P], or DUE-only [total, B, P] when the CUEs follow the traffic model
    actions = torch.empty((total, b, n_agents), dtype=torch.int32, device=dev)
    if n_agents == n and c:
        actions[:, :, :c] = torch.randint(0, r * pc, (total, b, c), generator=g, device=dev, dtype=torch.int32)
    if p:
        actions[:, :, n_agents - p:] = torch.randint(0, r * pd, (total, b, p), generator=g, device=dev, dtype=torch.int32)

    gatherer = None
    if use_dist and not args.no_gather:
        from gym_d2d_amd.distributed import StepGatherer
        gatherer = StepGatherer(b, n, dev)

    # An event pair around ONE launch adds about 3-6 us to what it measures (tools/probes/launch_floor.hip: a single
    # empty wave reads 6 us) and costs host time per step: irrelevant beside the 3.7 ms obs kernel, 10-30 % of a 17-40 us
    # step.  So: two kernels per step with the expansion dominant (LinearObs, N > 128) -> per-launch events inside the
    # timed region.  One kernel per step (compact table, or small N with the expansion fused) -> the timed region runs
    # without events, and a second pass over the same steps brackets GROUPS of back-to-back launches with one event pair
    # (on the stream the kernels run on): average launch duration = group time / launches, inter-launch gaps included.
    events_in_timed = args.obs == 'linear' and n > 128
    GROUP = 20

    def run(k0, k1):
        for k in range(k0, k1):
            new_episode = args.with_reset and k % 10 == 0
            if new_episode:                              # EPISODE_LENGTH = 10 (d2d_env.py:16): new layout per episode
                h.reset_positions(1234, k // 10)
            h.step(actions[k].data_ptr())
            if gatherer is not None and (new_episode or k == 0):
                gatherer.gather_positions(env._t['table'])      # position columns only change at reset (not per step)
            if gatherer is not None:
                gatherer.launch(env._t['reward'], env._t['table'])
        if gatherer is not None:
            gatherer.wait()

    def fence():
        if not stub:
            torch.cuda.synchronize(dev)
        if use_dist:
            dist.barrier()
        if not stub:
            torch.cuda.synchronize(dev)

    run(0, args.warmup)
    fence()
    h.profile_reset()
    h.profile_enable(events_in_timed)
    t0 = time.perf_counter()
    run(args.warmup, total)
    fence()
    dt = time.perf_counter() - t0
    step_ms, step_n = h.profile_read(0)
    obs_ms, obs_n = h.profile_read(1)
    h.profile_enable(False)
    if not events_in_timed and not stub:
        stream = torch.cuda.current_stream(dev)          # VecD2DEnv runs the library's kernels on this stream
        pairs = []
        for k0 in range(args.warmup, total, GROUP):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            k1 = min(k0 + GROUP, total)
            e0.record(stream)
            for k in range(k0, k1):
                h.step(actions[k].data_ptr())
            e1.record(stream)
            pairs.append((e0, e1, k1 - k0))
        torch.cuda.synchronize(dev)
        step_ms = sum(e0.elapsed_time(e1) for e0, e1, _ in pairs)
        step_n = sum(c for _, _, c in pairs)
        obs_ms, obs_n = 0.0, 0
    flags = env.status_flags()

    # report the 'core' mode beside the with-obs mode.  Same envs, same actions, obs_fn swapped for the
    # compact table (what a learner that builds its own features consumes): the step is then the step kernel alone.
    core = None
    if world == 1 and rank == 0 and not stub and args.obs == 'linear' and n > 128:
        h.set_obs_mode(_native.OBS_TABLE)
        for k in range(args.warmup):
            h.step(actions[k].data_ptr())
        torch.cuda.synchronize(dev)
        t0c = time.perf_counter()
        for k in range(args.warmup, total):
            h.step(actions[k].data_ptr())
        torch.cuda.synchronize(dev)
        dtc = time.perf_counter() - t0c
        stream = torch.cuda.current_stream(dev)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for k in range(args.warmup, total):
            h.step(actions[k].data_ptr())
        e1.record(stream)
        torch.cuda.synchronize(dev)
        k_ms = e0.elapsed_time(e1) / args.steps
        h.set_obs_mode(_native.OBS_LINEAR)
        core_bytes_launch = b * n * 64.0
        core = {'obs_mode': 'table (compact [B, N, 6] obs, no LinearObs expansion)', 'value': b * n * args.steps / dtc,
                'unit': 'agent-steps/s', 'ms_per_step': dtc / args.steps * 1e3,
                'roofline': {'kernel': 'step_kernel', 'bound': 'hbm', 'achieved': core_bytes_launch / (k_ms * 1e-3) / 1e9,
                             'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': core_bytes_launch / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                             'avg_launch_ms': k_ms, 'algorithmic_bytes_per_launch': core_bytes_launch,
                             'timing': f'one HIP event pair around {args.steps} back-to-back launches'}}

    dist_info = {}
    if use_dist:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        # evidence that the collective saw every rank: a count all-reduce, and the last step's rewards summed two ways
        # (all-reduce of the local sums vs the sum of what the all-gather delivered)
        ones = torch.ones(1, device=dev, dtype=torch.float64)
        dist.all_reduce(ones)
        local_sum = env._t['reward'][:, 0].double().sum().reshape(1)
        dist.all_reduce(local_sum)
        dist_info = {'rccl_ranks': dist.get_world_size(), 'backend': dist.get_backend(),
                     'allreduce_rank_count': float(ones.item()), 'allreduce_reward_checksum': float(local_sum.item())}
        if gatherer is not None:
            all_reward, all_signal = gatherer.wait()
            gsum = float(all_reward.double().sum().item())
            dist_info['allgather_reward_checksum'] = gsum
            dist_info['allgather_envs'] = int(all_reward.numel())
            dist_info['checksums_agree'] = bool(abs(gsum - dist_info['allreduce_reward_checksum'])
                                                <= 1e-6 * max(1.0, abs(gsum)))

    if rank == 0:
        agent_steps = b * n * args.steps * world
        value = agent_steps / dt
        core_bytes = 40.0                      # action 4 + positions 16 + outputs 16 + reward 4
        obs_bytes = 24.0 * n                   # LinearObs materialised: 6N floats per agent
        fused = args.obs == 'linear' and not events_in_timed   # small N: the expansion runs inside the step launch
        if fused:
            per_launch = b * n * (core_bytes + obs_bytes)
            avg_ms = step_ms / max(step_n, 1)
            roof = {'kernel': 'step_kernel (LinearObs expansion fused)'}
        elif args.obs == 'linear' and obs_n:
            # dominant kernel: obs expansion.  Algorithmic bytes per launch = B*N*(24N written + 24 read of T)
            per_launch = b * n * (obs_bytes + 24.0)
            avg_ms = obs_ms / obs_n
            roof = {'kernel': 'obs_expand_kernel'}
        else:
            per_launch = b * n * (core_bytes + (24.0 if args.obs == 'table' else 0.0))
            avg_ms = step_ms / max(step_n, 1)
            roof = {'kernel': 'step_kernel'}
        ach = per_launch / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        roof.update({'bound': 'hbm', 'achieved': ach, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': ach / HBM_PEAK_GBS,
                     'avg_launch_ms': avg_ms, 'algorithmic_bytes_per_launch': per_launch, 'traffic': None,
                     'timing': ('HIP events around every launch on the library stream, inside the timed region' if events_in_timed
                                else f'HIP events around groups of {GROUP} back-to-back launches (one kernel per step) on the '
                                     'library stream, in a second pass over the same steps; the timed region itself runs '
                                     'without events')})
        if not stub:
            attach_traffic(roof, args, w)
        out = {
            'metric': 'env agent-steps/sec (batch x agents)', 'value': value, 'unit': 'agent-steps/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': dt / args.steps * 1e3,
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': w['name'], 'envs_per_gpu': b, 'links_per_env': n, 'obs_mode': args.obs,
                       'reward_fn': 'SystemCapacity',
                       'path_loss': 'FreeSpacePathLoss (plugin class)' if w.get('plugin') else 'LogDistance(ple=2)',
                       'positions': 'redrawn on the device every 10 steps' if args.with_reset else 'fixed over the run',
                       'actions': 'fresh i.i.d. per step (pre-generated in HBM)',
                       'cue_actions': cue_mode + (' (UplinkTrafficModel round-robin, held in the kernel\'s link records; agents supply DUE actions only)'
                                                  if cue_mode == 'traffic' else ' (agents supply CUE and DUE actions)'),
                       'parallelism': f'env-shard x{world}' + (' + per-step all-gather(reward, sinr/snr columns of the obs table; position columns once per episode)'
                                                                     if gatherer else '')},
            'roofline': roof,
            'kernels': {'step_kernel_ms': step_ms / max(step_n, 1), 'obs_expand_kernel_ms': (obs_ms / obs_n) if obs_n else None},
            'algorithmic_bytes_per_agent_step': core_bytes + (obs_bytes if args.obs == 'linear' else (24.0 if args.obs == 'table' else 0.0)),
            'status_flags': flags,
            'target_agent_steps_per_s': 1e7,
        }
        out.update(dist_info)
        if stub:
            out['stub'] = 'CPU plumbing test - not a measurement'
        if args.share_gpu:
            out['shared_gpu'] = f'{world} ranks on ONE GPU over gloo: multi-rank correctness test - not a measurement'
        if single_ms is not None:
            out['single_env_step_ms'] = single_ms
        if core is not None:
            out['core_mode'] = core
        if cpu is not None:
            out['cpu_baseline'] = cpu
        sys.stdout.flush()
        os.dup2(saved_stdout, 1)                 # the real stdout is back for the one line that belongs there
        print(json.dumps(out), flush=True)
        os.dup2(2, 1)                            # teardown chatter (RCCL / ROCm) goes to stderr again
    env.close()
    if use_dist:
        dist.destroy_process_group()


def attach_traffic(roof, args, w):
    """HBM bytes per launch come from rocprofv3 PMC passes, which cannot be collected from inside this process: the
    figure is taken from the newest committed summary for this kernel and workload, but only if that summary was
    made from the SAME kernel sources that are running now (digest of csrc/ recorded by tools/summarize_profiles.py);
    otherwise it stays null and the mismatch is reported."""
    if args.envs:
        return
    from gym_d2d_amd.build import source_digest
    digest = source_digest()
    for path in sorted((ROOT / 'profiles').glob('r*_pmc_*.json'), reverse=True):
        try:
            rec = json.loads(path.read_text())
        except Exception:
            continue
        if rec.get('workload_key') != f'{args.workload}/{args.obs}':
            continue
        for kname, d in rec.get('kernels', {}).items():
            if roof['kernel'].split(' ')[0] in kname and 'hbm_bytes_per_launch' in d:
                if rec.get('source_digest') == digest:
                    roof['traffic'] = d['hbm_bytes_per_launch']
                    roof['traffic_source'] = (f'profiles/{path.name} (WRITE_SIZE + 2*FETCH_SIZE, separate --pmc passes; '
                                              f'kernel sources {digest[:12]} = the ones running)')
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


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    args = parse_args(argv)
    if args.gpus > 1 and 'RANK' not in os.environ:
        sys.exit(launch_ranks(args.gpus, argv))
    worker(args)


if __name__ == '__main__':
    main()
