#!/usr/bin/env python3
"""Benchmark of the hot path: D2DEnv.step for a batch of environments on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload stress|default|plugin] [--obs linear|table|none]

A "step" is one pass of the fused path over the whole batch: action decode -> SINR / SNR / rate / capacity -> reward
-> observation table -> LinearObs expansion (reference semantics: obs materialised as [B, N, 6N] in HBM), with fresh
i.i.d. actions every step (pre-generated in HBM; positions fixed for the run - the reference's step never moves
devices, simulator.py:61-75).  Metric: agent-steps/s = B * N * steps / wall seconds (whole job, all GPUs).

N > 1: one process per GPU (torch.distributed / RCCL), env axis sharded 4096 per GPU (weak scaling), one all-gather
per step of rewards + the per-step columns of the compact obs table, overlapped on a side stream.

Prints ONE JSON line on rank 0 with `roofline` (dominant kernel, HIP-event timed on the library's stream) and
`cpu_baseline` (the NumPy fp64 oracle timed on this box's host cores on a bounded sample; rank 0, N = 1 only).
"""
import argparse
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

import numpy as np
import torch

WORKLOADS = {
    # BASELINE.json configs[2]: the configuration the target metric is quoted on
    'stress': dict(name='4096 envs x (256 CUE + 256 DUE pairs, 256 RB), LogDistance', envs=4096, rbs=256, cues=256, dues=256),
    # BASELINE.json configs[1]
    'default': dict(name='1024 envs x (25 CUE + 25 DUE pairs, 25 RB), LogDistance', envs=1024, rbs=25, cues=25, dues=25),
    # BASELINE.json configs[3]: the plugin-ABI swap - FreeSpacePathLoss class through env_config['path_loss_model'] and
    # a custom array ObsFunction (own link only); same sizes as 'stress'
    'plugin': dict(name='4096 envs x (256 CUE + 256 DUE pairs, 256 RB), FreeSpacePathLoss + OwnLinkObsFunction plugins',
                   envs=4096, rbs=256, cues=256, dues=256, plugin=True),
}
HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def cpu_baseline(w, seconds_budget=20.0):
    """Time the NumPy fp64 oracle (a port of the reference's algorithm; the Python reference cannot travel to this
    box) on a bounded sample of the same workload, including the materialised LinearObs."""
    from oracle import d2d_oracle as orc
    sys.path.insert(0, str(ROOT / 'tests'))
    from sim_util import default_links, random_layout
    rng = np.random.default_rng(1234)
    c, p, r = w['cues'], w['dues'], w['rbs']
    n = c + p
    ids, cfgs, is_bs = orc.device_configs(c, p)
    cols = orc.device_columns(cfgs, is_bs)
    tx, rx, ty = default_links(c, p)
    envs_per_call = 16 if n > 100 else 256
    pos = random_layout(rng, envs_per_call, c, p).astype(np.float64)
    done, t0 = 0, time.perf_counter()
    while True:
        raw = np.concatenate([rng.integers(0, r * 24, (envs_per_call, c)), rng.integers(0, r * 21, (envs_per_call, p))], 1)
        orc.full_step(pos, tx, rx, ty, raw, cols, orc.PathLossSpec(), with_obs=True, chunk=16)
        done += envs_per_call
        dt = time.perf_counter() - t0
        if dt > seconds_budget * 0.5 or done >= 4096:
            break
    return {'value': done * n / dt, 'unit': 'agent-steps/s', 'cores': 1, 'kind': 'port',
            'sample': f'{done} env-steps of the same workload ({done * n} agent-steps, obs materialised), '
                      f'NumPy fp64 oracle, single thread, {dt:.1f} s; host has {os.cpu_count()} cores',
            'reference_pure_python_1core_build_container': 6.3e3 if n > 100 else 3.3e4}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=100)
    ap.add_argument('--warmup', type=int, default=10)
    ap.add_argument('--workload', default='stress', choices=sorted(WORKLOADS))
    ap.add_argument('--obs', default='linear', choices=['linear', 'table', 'none'])
    ap.add_argument('--envs', type=int, default=0, help='override envs per GPU')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-gather', action='store_true', help='N > 1: skip the per-step all-gather')
    ap.add_argument('--with-reset', action='store_true', help='redraw all device positions every 10 steps (device-side reset)')
    ap.add_argument('--force-dist', action='store_true', help='init RCCL and run the gather path even with one rank (test hook)')
    ap.add_argument('--tune', default='', help='comma list key=value: rows,nt,xcd,bucket')
    args = ap.parse_args()

    rank = int(os.environ.get('RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    local = int(os.environ.get('LOCAL_RANK', 0))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit('launch with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...')
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    # stdout carries exactly ONE line (the JSON): RCCL prints a version banner to fd 1 when its first communicator is
    # created, so everything until the result is ready goes to stderr instead
    sys.stdout.flush()
    saved_stdout = os.dup(1)
    os.dup2(2, 1)
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    use_dist = world > 1 or args.force_dist
    if use_dist:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29511')
        dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)

    from gym_d2d_amd import _native
    from gym_d2d_amd.envs import VecD2DEnv
    from gym_d2d_amd.envs.obs_fn import LinearObsFunction, OwnLinkObsFunction

    w = dict(WORKLOADS[args.workload])
    if args.envs:
        w['envs'] = args.envs
    b, c, p, r = w['envs'], w['cues'], w['dues'], w['rbs']
    n = c + p
    if w.get('plugin'):
        args.obs = 'table'
    cfg = {'num_rbs': r, 'num_cues': c, 'num_due_pairs': p, 'device_ordinal': local,
           'obs_fn': LinearObsFunction if args.obs == 'linear' else OwnLinkObsFunction}
    if w.get('plugin'):
        from gym_d2d_amd.path_loss import FreeSpacePathLoss
        cfg['path_loss_model'] = FreeSpacePathLoss
    env = VecD2DEnv(cfg, num_envs=b, first_env=rank * b)
    h = env.simulator.handle
    if args.obs == 'none':
        h.set_obs_mode(_native.OBS_NONE)
    for kv in filter(None, args.tune.split(',')):
        k, v = kv.split('=')
        if k == 'bucket':
            h.set_bucketing(bool(int(v)))
        else:
            h.set_tuning({'rows': _native.TUNE_OBS_ROWS_PER_WG, 'nt': _native.TUNE_OBS_NONTEMPORAL,
                          'xcd': _native.TUNE_OBS_XCD_REMAP}[k], int(v))
    env.reset(seed=1234)

    total = args.steps + args.warmup
    g = torch.Generator(device=dev)
    g.manual_seed(1234 + rank)
    pc, pd = env.num_pwr_actions['cue'], env.num_pwr_actions['due']
    actions = torch.empty((total, b, n), dtype=torch.int32, device=dev)
    if c:
        actions[:, :, :c] = torch.randint(0, r * pc, (total, b, c), generator=g, device=dev, dtype=torch.int32)
    if p:
        actions[:, :, c:] = torch.randint(0, r * pd, (total, b, p), generator=g, device=dev, dtype=torch.int32)

    gatherer = None
    if use_dist and not args.no_gather:
        from gym_d2d_amd.distributed import StepGatherer
        gatherer = StepGatherer(b, n, dev)

    def run(k0, k1):
        for k in range(k0, k1):
            new_episode = args.with_reset and k % 10 == 0
            if new_episode:                              # EPISODE_LENGTH = 10 (d2d_env.py:16): new layout per episode
                h.reset_positions(1234, k // 10)
            h.step(actions[k].data_ptr())
            if gatherer is not None and (new_episode or k == k0):
                gatherer.gather_positions(env._t['table'])      # position columns only change at reset
            if gatherer is not None:
                gatherer.launch(env._t['reward'], env._t['table'])
        if gatherer is not None:
            gatherer.wait()

    def fence():
        torch.cuda.synchronize(dev)
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize(dev)

    run(0, args.warmup)
    fence()
    h.profile_reset()
    h.profile_enable(True)
    t0 = time.perf_counter()
    run(args.warmup, total)
    fence()
    dt = time.perf_counter() - t0
    step_ms, step_n = h.profile_read(0)
    obs_ms, obs_n = h.profile_read(1)
    h.profile_enable(False)
    flags = env.status_flags()

    if use_dist:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    if rank == 0:
        agent_steps = b * n * args.steps * world
        value = agent_steps / dt
        core_bytes = 40.0                      # action 4 + positions 16 + outputs 16 + reward 4
        obs_bytes = 24.0 * n                   # LinearObs materialised: 6N floats per agent
        if args.obs == 'linear' and obs_n:
            # dominant kernel: obs expansion.  Algorithmic bytes per launch = B*N*(24N written + 24 read of T)
            per_launch = b * n * (obs_bytes + 24.0)
            avg_s = obs_ms / obs_n * 1e-3
            roof = {'kernel': 'obs_expand_kernel', 'bound': 'hbm', 'achieved': per_launch / avg_s / 1e9,
                    'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'avg_launch_ms': obs_ms / obs_n,
                    'algorithmic_bytes_per_launch': per_launch, 'traffic': None}
        else:
            per_launch = b * n * (core_bytes + (24.0 if args.obs == 'table' else 0.0))
            avg_s = step_ms / max(step_n, 1) * 1e-3
            roof = {'kernel': 'step_kernel', 'bound': 'hbm', 'achieved': per_launch / avg_s / 1e9,
                    'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'avg_launch_ms': step_ms / max(step_n, 1),
                    'algorithmic_bytes_per_launch': per_launch, 'traffic': None}
        roof['frac'] = roof['achieved'] / roof['peak']
        # HBM bytes per launch from rocprofv3 PMC passes (cannot be collected from inside this process): taken from
        # the committed summary of the same workload, when one exists for this kernel
        pmc = sorted((ROOT / 'profiles').glob('r*_pmc_hbm_traffic.json'))
        if pmc and args.workload == 'stress' and not args.envs:
            rec = json.loads(pmc[-1].read_text())
            for kname, d in rec['kernels'].items():
                if roof['kernel'] in kname:
                    roof['traffic'] = d['hbm_bytes_per_launch']
                    roof['traffic_source'] = f'profiles/{pmc[-1].name} (WRITE_SIZE + 2*FETCH_SIZE, separate --pmc passes)'
        out = {
            'metric': 'env agent-steps/sec (batch x agents)', 'value': value, 'unit': 'agent-steps/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': dt / args.steps * 1e3,
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': w['name'], 'envs_per_gpu': b, 'links_per_env': n, 'obs_mode': args.obs,
                       'reward_fn': 'SystemCapacity',
                       'path_loss': 'FreeSpacePathLoss (plugin class)' if w.get('plugin') else 'LogDistance(ple=2)',
                       'positions': 'redrawn on the device every 10 steps' if args.with_reset else 'fixed over the run',
                       'actions': 'fresh i.i.d. per step (pre-generated in HBM)',
                       'parallelism': f'env-shard x{world}' + (' + per-step all-gather(reward, sinr/snr columns of the obs table; position columns once per episode)'
                                                                     if gatherer else '')},
            'roofline': roof,
            'kernels': {'step_kernel_ms': step_ms / max(step_n, 1), 'obs_expand_kernel_ms': (obs_ms / obs_n) if obs_n else None},
            'algorithmic_bytes_per_agent_step': core_bytes + (obs_bytes if args.obs == 'linear' else (24.0 if args.obs == 'table' else 0.0)),
            'status_flags': flags,
            'target_agent_steps_per_s': 1e7,
        }
        if world == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline(w)
        sys.stdout.flush()
        os.dup2(saved_stdout, 1)                 # the real stdout is back for the one line that belongs there
        print(json.dumps(out), flush=True)
        os.dup2(2, 1)                            # teardown chatter (RCCL / ROCm) goes to stderr again
    if use_dist:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
