"""Two - and EIGHT, BASELINE config 5's rank count - RANKS, each stepping its own env shard with the HIP library, gathered by
StepGatherer on the GPU - on ONE device: all processes use cuda:0 and the gloo backend carries the collectives (RCCL refuses two ranks on one GPU, and the GPU box
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
B_GLOBAL, STEPS = 48, 3


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _actions(step, first, count):
    rng = np.random.default_rng(1000 + step)
    whole = np.concatenate([rng.integers(0, 6 * 24, (B_GLOBAL, 5)), rng.integers(0, 6 * 21, (B_GLOBAL, 9))], 1)
    return whole[first:first + count].astype(np.int32)


def _rank(rank, world, port, out_dir):
    for p in (str(ROOT), str(ROOT / 'tests')):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    import torch
    import torch.distributed as dist
    from gym_d2d_amd.distributed import StepGatherer, expand_table, shard_range
    from gym_d2d_amd.envs import VecD2DEnv
    dist.init_process_group('gloo', rank=rank, world_size=world)
    first, end = shard_range(B_GLOBAL, world, rank)
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
    # the compact-obs plan on the same streams: rewards only, and the observation columns on every 2nd launch
    g2 = StepGatherer(end - first, 14, dev, mode='rewards')
    g3 = StepGatherer(end - first, 14, dev, signal_every=2)
    g3.gather_positions(env._t['table'])
    for k in range(STEPS):
        env.step(torch.as_tensor(_actions(k, first, end - first), device=dev))
        g2.launch(env._t['reward']); g3.launch(env._t['reward'], env._t['table'])
        r2, s2 = g2.wait()
        r3, s3 = g3.wait()
        torch.cuda.synchronize()
        assert s2 is None and np.array_equal(r2.cpu().numpy(), outs[k][0]) and np.array_equal(r3.cpu().numpy(), outs[k][0])
        assert g3.signal_step == (k // 2) * 2 and np.array_equal(g3.table().cpu().numpy(), outs[(k // 2) * 2][1])
    # rewards in a device-side ring, gathered every 3rd launch
    g4 = StepGatherer(end - first, 14, dev, mode='rewards', reward_every=3)
    for k in range(STEPS):
        env.step(torch.as_tensor(_actions(k, first, end - first), device=dev))
        g4.launch(env._t['reward'])
    block, _ = g4.wait()
    torch.cuda.synchronize()
    assert g4.reward_step == 0 and np.array_equal(block.cpu().numpy(), np.stack([outs[k][0] for k in range(STEPS)]))
    # the learner configuration: every rank runs D2D_OBS_NONE with the per-env reward and no decoded planes;
    # the planes plan gathers the (sinr, snr) planes and the library's link-position rows - the same global table, bit for bit
    from gym_d2d_amd.envs.obs_fn import SignalPlanesObsFunction
    lean = VecD2DEnv(dict(CFG, obs_fn=SignalPlanesObsFunction), num_envs=end - first, first_env=first, export_actions=False,
                     reward_per_env=True)
    lean.reset(seed=99)
    g5 = StepGatherer(end - first, 14, dev, mode='planes')
    g5.gather_positions(lean.link_positions())
    for k in range(STEPS):
        (sinr, snr), rew_env, _, _ = lean.step(torch.as_tensor(_actions(k, first, end - first), device=dev))
        g5.launch(rew_env, sinr=sinr, snr=snr)
        r5, planes = g5.wait()
        t5 = g5.table()
        torch.cuda.synchronize()
        assert np.array_equal(r5.cpu().numpy(), outs[k][0]) and np.array_equal(t5.cpu().numpy(), outs[k][1]), k
        assert tuple(planes[0].shape) == (B_GLOBAL, 14)
    lean.close()
    if rank == 0:
        np.savez(Path(out_dir) / 'gathered.npz', **{f'{name}{k}': arr for k, o in enumerate(outs)
                                                    for name, arr in zip(('reward', 'table', 'obs'), o)})
    dist.barrier()
    env.close()
    dist.destroy_process_group()


@pytest.mark.parametrize('world', [2, 8])
def test_ranks_sharing_one_gpu_reproduce_the_single_process_batch(tmp_path, world):
    import torch
    import torch.multiprocessing as mp
    from gym_d2d_amd.envs import VecD2DEnv
    mp.spawn(_rank, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    got = np.load(tmp_path / 'gathered.npz')
    env = VecD2DEnv(dict(CFG), num_envs=B_GLOBAL)
    env.reset(seed=99)
    for k in range(STEPS):
        obs, rew, dones, info = env.step(torch.as_tensor(_actions(k, 0, B_GLOBAL), device=env.device))
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
    # the fields that make a first real SCALE line self-explaining
    assert abs(two['value_per_gpu'] * 2 - two['value']) < 1e-6 * two['value'] and len(two['per_rank']) == 2
    g = two['gather']
    assert g['mode'] == 'table' and g['bytes_per_gpu_per_step'] == 64 * 4 + 64 * 50 * 2 * 4 and g['gather_ms_per_step'] > 0
    assert g['ms_per_step_without_gather'] > 0 and 'gather_exposed_ms' in g
    assert all(r['step_kernel_ms'] > 0 for r in two['per_rank'])
    one = subprocess.run([sys.executable, str(ROOT / 'bench.py'), '--gpus', '1', '--force-dist', '--workload', 'default',
                          '--obs', 'table', '--envs', '64', '--steps', '4', '--warmup', '1', '--no-cpu-baseline', '--no-single-env-latency'],
                         capture_output=True, text=True, timeout=600,
                         env={k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK')})
    assert one.returncode == 0, one.stderr[-3000:]
    rank0 = json.loads([ln for ln in one.stdout.splitlines() if ln.strip()][0])
    assert abs(two['allgather_reward_checksum'] - 2 * rank0['allgather_reward_checksum']) > 1e-3


def test_bench_two_ranks_headline_shape_with_core_mode_gathers():
    """The driver's N > 1 command line in miniature: the stress workload's shape (512 links, LinearObs: two kernels per step) with
    two ranks sharing the GPU, so that `core_mode` runs WITH its gatherers - the rewards ring for the table entries and the planes
    plan for `planes_only` - which no one-rank run reaches."""
    import json
    import subprocess
    cmd = [sys.executable, str(ROOT / 'bench.py'), '--gpus', '2', '--share-gpu', '--envs', '16', '--steps', '8', '--warmup', '2',
           '--no-cpu-baseline', '--no-single-env-latency']
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900,
                       env={k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK')})
    assert r.returncode == 0, r.stderr[-3000:]
    two = json.loads([ln for ln in r.stdout.splitlines() if ln.strip()][0])
    assert two['n_gpus'] == 2 and two['checksums_agree'] is True and two['gather']['mode'] == 'table'
    core = two['core_mode']
    assert core['gather']['mode'] == 'rewards' and core['value'] > 0 and core['with_decoded_rb_pwr_export']['value'] > 0
    po = core['planes_only']
    assert po['gather']['mode'] == 'planes' and po['gather']['bytes_per_gpu_per_step'] == 16 * 4 + 16 * 512 * 8 and po['value'] > 0
    assert abs(po['algorithmic_bytes_per_agent_step'] - (36.0 + 4.0 / 512)) < 1e-9
    assert two['cpu_baseline'].startswith('N=1 line only')


def test_bench_two_ranks_obs_less_learner_configuration():
    """`bench.py --gpus 2 --share-gpu --obs none --reward-per-env --no-export`: every rank runs D2D_OBS_NONE through the public API and the
    per-step gather is the planes plan (chosen by itself: no table exists); the collectives saw both ranks and the gathered per-env
    rewards sum to the all-reduced local sums."""
    import json
    import subprocess
    cmd = [sys.executable, str(ROOT / 'bench.py'), '--gpus', '2', '--share-gpu', '--workload', 'default', '--obs', 'none', '--reward-per-env',
           '--no-export', '--envs', '64', '--steps', '4', '--warmup', '1', '--no-cpu-baseline', '--no-single-env-latency']
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600,
                       env={k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK')})
    assert r.returncode == 0, r.stderr[-3000:]
    two = json.loads([ln for ln in r.stdout.splitlines() if ln.strip()][0])
    assert two['n_gpus'] == 2 and two['allgather_envs'] == 128 and two['checksums_agree'] is True
    g = two['gather']
    assert g['mode'] == 'planes' and g['bytes_per_gpu_per_step'] == 64 * 4 + 64 * 50 * 2 * 4
    assert two['config']['obs_mode'] == 'none' and two['algorithmic_bytes_per_agent_step'] == 36.0 + 4.0 / 50


def test_bench_native_gather_backend_with_one_rank():
    """`bench.py --gather-backend native`: the per-step gather through the C ABI's own RCCL entry (d2d_comm_init /
    d2d_allgather on the side stream) instead of torch.distributed - with one rank, which is all RCCL admits on a one-GPU box;
    the headline gather (table plan) and core_mode's rewards-only plan each build their own communicator."""
    import json
    import subprocess
    cmd = [sys.executable, str(ROOT / 'bench.py'), '--gpus', '1', '--force-dist', '--gather-backend', 'native', '--envs', '64',
           '--steps', '4', '--warmup', '1', '--reward-every', '2', '--no-cpu-baseline', '--no-single-env-latency']
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600,
                       env={k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK')})
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.strip()][0])
    assert line['rccl_ranks'] == 1 and line['checksums_agree'] is True and line['gather']['backend'] == 'native'
    assert line['allgather_envs'] == 64 and line['gather']['gather_ms_per_step'] > 0
    assert line['core_mode']['gather']['mode'] == 'rewards' and line['core_mode']['gather']['reward_every'] == 2
    assert line['core_mode']['gather']['gather_ms_per_step'] > 0
