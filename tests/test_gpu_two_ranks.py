"""Two RANKS, each stepping its own env shard with the HIP library, gathered by StepGatherer on the GPU - on ONE device:
both processes use cuda:0 and the gloo backend carries the collectives (RCCL refuses two ranks on one GPU, and the GPU box
has one).  What this exercises that the CPU gloo test cannot: real d2d_step outputs per rank (first_env offsets, counter-
based reset), the CUDA side-stream / event choreography of StepGatherer, and the consumer-side d2d_expand_table - all
compared with ONE process stepping the whole batch."""
import os
import socket
import sys
from pathlib import Path

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent
CFG = {'num_rbs': 6, 'num_cues': 5, 'num_due_pairs': 9}
B_LOCAL, WORLD, STEPS = 24, 2, 3


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _actions(step, first, count):
    rng = np.random.default_rng(1000 + step)
    whole = np.concatenate([rng.integers(0, 6 * 24, (B_LOCAL * WORLD, 5)), rng.integers(0, 6 * 21, (B_LOCAL * WORLD, 9))], 1)
    return whole[first:first + count].astype(np.int32)


def _rank(rank, port, out_dir):
    for p in (str(ROOT), str(ROOT / 'tests')):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    import torch
    import torch.distributed as dist
    from gym_d2d_amd.distributed import StepGatherer, expand_table, shard_range
    from gym_d2d_amd.envs import VecD2DEnv
    dist.init_process_group('gloo', rank=rank, world_size=WORLD)
    first, end = shard_range(B_LOCAL * WORLD, WORLD, rank)
    env = VecD2DEnv(dict(CFG), num_envs=end - first, first_env=first)
    dev = env.device
    obs = env.reset(seed=99)
    g = StepGatherer(end - first, 14, dev)
    g.gather_positions(env._t['table'])
    outs = []
    for k in range(STEPS):
        obs, rew, dones, info = env.step(torch.as_tensor(_actions(k, first, end - first), device=dev))
        g.launch(env._t['reward'], env._t['table'])
        rewards, signal = g.wait()
        table = g.table()
        torch.cuda.synchronize()
        outs.append((rewards.cpu().numpy().copy(), table.cpu().numpy().copy(),
                     expand_table(table, env.simulator.handle).cpu().numpy().copy()))
    if rank == 0:
        np.savez(Path(out_dir) / 'gathered.npz', **{f'{name}{k}': arr for k, o in enumerate(outs)
                                                    for name, arr in zip(('reward', 'table', 'obs'), o)})
    dist.barrier()
    env.close()
    dist.destroy_process_group()


def test_two_ranks_sharing_one_gpu_reproduce_the_single_process_batch(tmp_path):
    import torch
    import torch.multiprocessing as mp
    from gym_d2d_amd.envs import VecD2DEnv
    mp.spawn(_rank, args=(_free_port(), str(tmp_path)), nprocs=WORLD, join=True)
    got = np.load(tmp_path / 'gathered.npz')
    env = VecD2DEnv(dict(CFG), num_envs=B_LOCAL * WORLD)
    env.reset(seed=99)
    for k in range(STEPS):
        obs, rew, dones, info = env.step(torch.as_tensor(_actions(k, 0, B_LOCAL * WORLD), device=env.device))
        torch.cuda.synchronize()
        # sharding changes nothing: positions and reset actions are keyed by global env index, steps are per env
        assert np.array_equal(got[f'reward{k}'], rew[:, 0].cpu().numpy()), k
        assert np.array_equal(got[f'table{k}'], env._t['table'].cpu().numpy()), k
        assert np.array_equal(got[f'obs{k}'], obs.cpu().numpy()), k
    env.close()


def test_bench_two_ranks_on_one_gpu_agree_with_one_rank_of_twice_the_envs():
    """bench.py's real worker with N = 2 (self-launched, both ranks on cuda:0, gloo): the collectives saw 2 ranks, the
    all-gathered rewards sum to the all-reduced local sums, and that sum is NOT rank-0's sum twice (the shards differ:
    first_env offsets reach the reset sampler).  The driver's 8-GPU run is this code with RCCL as the transport."""
    import json
    import subprocess
    cmd = [sys.executable, str(ROOT / 'bench.py'), '--gpus', '2', '--share-gpu', '--workload', 'default', '--obs', 'table',
           '--envs', '64', '--steps', '4', '--warmup', '1', '--no-cpu-baseline', '--no-single-env-latency']
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600,
                       env={k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK')})
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout
    two = json.loads(lines[0])
    assert two['n_gpus'] == 2 and two['rccl_ranks'] == 2 and two['allreduce_rank_count'] == 2.0
    assert two['allgather_envs'] == 128 and two['checksums_agree'] is True and 'shared_gpu' in two
    assert two['status_flags'] == 0
    one = subprocess.run([sys.executable, str(ROOT / 'bench.py'), '--gpus', '1', '--force-dist', '--workload', 'default',
                          '--obs', 'table', '--envs', '64', '--steps', '4', '--warmup', '1', '--no-cpu-baseline', '--no-single-env-latency'],
                         capture_output=True, text=True, timeout=600,
                         env={k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK')})
    assert one.returncode == 0, one.stderr[-3000:]
    rank0 = json.loads([ln for ln in one.stdout.splitlines() if ln.strip()][0])
    assert abs(two['allgather_reward_checksum'] - 2 * rank0['allgather_reward_checksum']) > 1e-3
