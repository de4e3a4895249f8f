"""World-size-2 test of the multi-GPU path on CPU (gloo): shard ranges, rank-major gather order, consumer-side
expansion.  The per-GPU step itself needs a GPU (covered by -m gpu); here local results are synthesised by the
oracle so that the concatenation can be compared with a single-process run."""
import os
import socket
import sys
from pathlib import Path

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = Path(__file__).resolve().parent.parent
for p in (str(ROOT), str(ROOT / 'tests')):
    if p not in sys.path:
        sys.path.insert(0, p)


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _oracle_results(global_envs, begin, end, c=3, p=4, r=3):
    """Deterministic per-env synthetic step results for envs [begin, end) of a global batch."""
    from oracle import d2d_oracle as orc
    from sim_util import default_links
    d = 1 + c + 2 * p
    u = orc.reset_uniforms(seed=42, episode=0, num_envs=global_envs, num_devices=d, tries=32)[begin:end]
    pos, _ = orc.sample_positions_from_uniforms(u, c, p)
    rng = np.random.default_rng(7)
    raw_all = np.concatenate([rng.integers(0, r * 24, (global_envs, c)), rng.integers(0, r * 21, (global_envs, p))], 1)
    tx, rx, ty = default_links(c, p)
    ids, cfgs, is_bs = orc.device_configs(c, p)
    st = orc.full_step(pos, tx, rx, ty, raw_all[begin:end], orc.device_columns(cfgs, is_bs), orc.PathLossSpec())
    n = c + p
    return (np.repeat(st['reward'][:, None], n, 1).astype(np.float32), st['table'].astype(np.float32),
            st['obs'].astype(np.float32))


def _worker(rank, world, port, global_envs, out_dir):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from gym_d2d_amd.distributed import StepGatherer, expand_table, shard_range
    begin, end = shard_range(global_envs, world, rank)
    reward, table, obs = _oracle_results(global_envs, begin, end)
    g = StepGatherer(end - begin, table.shape[1], torch.device('cpu'))
    t_table = torch.from_numpy(table)
    g.gather_positions(t_table)                          # once per episode: the four position columns
    for _ in range(2):                                   # twice: staging buffers are reusable
        g.launch(torch.from_numpy(reward), t_table)      # per step: rewards + (sinr, snr) only
        all_reward, all_signal = g.wait()
    assert tuple(all_signal.shape) == (global_envs, table.shape[1], 2)
    all_table = g.table()
    if rank == 0:
        np.save(Path(out_dir) / 'reward.npy', all_reward.numpy())
        np.save(Path(out_dir) / 'table.npy', all_table.numpy())
        np.save(Path(out_dir) / 'obs.npy', expand_table(all_table).numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_shard_range_partitions_the_batch():
    from gym_d2d_amd.distributed import shard_range
    for total, world in ((32768, 8), (10, 4), (7, 2), (3, 8)):
        spans = [shard_range(total, world, r) for r in range(world)]
        assert spans[0][0] == 0 and spans[-1][1] == total
        assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
        sizes = [e - b for b, e in spans]
        assert max(sizes) - min(sizes) <= 1
    assert shard_range(32768, 8, 3) == (12288, 16384)
    with pytest.raises(ValueError):
        shard_range(8, 2, 2)


@pytest.mark.parametrize('world,global_envs', [(2, 12), (8, 16)])
def test_multi_rank_gather_matches_single_process(tmp_path, world, global_envs):
    """World sizes 2 and 8 (BASELINE config 5's rank count) over gloo."""
    mp.spawn(_worker, args=(world, _free_port(), global_envs, str(tmp_path)), nprocs=world, join=True)
    reward, table, obs = _oracle_results(global_envs, 0, global_envs)
    assert np.array_equal(np.load(tmp_path / 'reward.npy'), reward[:, 0])
    assert np.array_equal(np.load(tmp_path / 'table.npy'), table)
    # the learner-side expansion of the gathered compact table == the expanded obs every rank holds locally
    assert np.array_equal(np.load(tmp_path / 'obs.npy'), obs)


def _modes_worker(rank, world, port, out_dir):
    """mode='rewards' (no observation columns travel) and signal_every=K (they travel on every K-th launch), and
    per-agent rewards: every rank checks what it received against what every rank is known to have sent."""
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from gym_d2d_amd.distributed import StepGatherer
    b, n = 3, 5
    dev = torch.device('cpu')

    def local(r, step):                                 # what rank r's step `step` produced
        reward = torch.arange(b * n, dtype=torch.float32).reshape(b, n) + 100.0 * r + 1000.0 * step
        table = torch.full((b, n, 6), float(r)) + 0.5 * step
        return reward, table
    errors = []
    g = StepGatherer(b, n, dev, mode='rewards')
    for step in range(3):
        g.launch(*local(rank, step))
        rew, sig = g.wait()
        want = torch.cat([local(r, step)[0][:, 0] for r in range(world)])
        if sig is not None or not torch.equal(rew, want):
            errors.append(f'rewards mode step {step}')
    if g.bytes_per_launch != b * 4 or g.bytes_per_signal_launch != b * 4:
        errors.append('rewards mode byte count')
    try:
        g.table()
        errors.append('rewards mode table() did not raise')
    except ValueError:
        pass
    g = StepGatherer(b, n, dev, signal_every=3, per_agent_reward=True)
    for step in range(7):
        g.launch(*local(rank, step))
        rew, sig = g.wait()
        want = torch.cat([local(r, step)[0] for r in range(world)])
        last = (step // 3) * 3                          # launches 0, 3, 6 carry the signal
        want_sig = torch.cat([local(r, last)[1][:, :, 4:6] for r in range(world)])
        if not torch.equal(rew, want) or not torch.equal(sig, want_sig) or g.signal_step != last:
            errors.append(f'signal_every step {step}')
    # rewards ride in a device-side ring and travel every 3rd launch: [3, B_global] blocks, rank-major along the env axis
    g = StepGatherer(b, n, dev, mode='rewards', reward_every=3)
    for step in range(7):
        g.launch(*local(rank, step))
        if step % 3 == 2:
            block, sig = g.wait()
            want = torch.stack([torch.cat([local(r, s)[0][:, 0] for r in range(world)]) for s in range(step - 2, step + 1)])
            if sig is not None or g.reward_step != step - 2 or not torch.equal(block, want):
                errors.append(f'reward ring block ending at step {step}')
    # a rollout that ends inside a block: flush() gathers the partly filled ring and says how many rows are valid
    g = StepGatherer(b, n, dev, mode='rewards', reward_every=4)
    for step in range(6):
        g.launch(*local(rank, step))
    valid = g.flush()
    block, _ = g.wait()
    want = torch.stack([torch.cat([local(r, s)[0][:, 0] for r in range(world)]) for s in (4, 5)])
    if valid != 2 or g.reward_step != 4 or tuple(block.shape)[0] != 2 or g.reward_valid != 2 or not torch.equal(block, want) or g.flush() != 0:
        errors.append('flush of a partial reward ring')
    for step in range(6, 10):                            # the next block starts clean after a flush
        g.launch(*local(rank, step))
    block, _ = g.wait()
    want = torch.stack([torch.cat([local(r, s)[0][:, 0] for r in range(world)]) for s in range(6, 10)])
    if g.reward_step != 6 or not torch.equal(block, want):
        errors.append('reward ring block after a flush')
    # mode 'planes': the (sinr, snr) planes and the link-position rows travel instead of the table; per-env reward vector;
    # the assembled table is bit-identical to the table plan's
    gt = StepGatherer(b, n, dev)
    gp = StepGatherer(b, n, dev, mode='planes')
    reward, table = local(rank, 5)
    table = table + torch.arange(b * n * 6, dtype=torch.float32).reshape(b, n, 6)
    gt.gather_positions(table)
    gp.gather_positions(table[:, :, :4].contiguous())    # D2D_BUF_LINK_POS rows
    gt.launch(reward, table)
    gp.launch(reward[:, 0].contiguous(), sinr=table[:, :, 4].contiguous(), snr=table[:, :, 5].contiguous())
    rt, _ = gt.wait()
    rp, planes = gp.wait()
    if not torch.equal(rt, rp) or not torch.equal(gt.table(), gp.table()) or tuple(planes[0].shape) != (world * b, n):
        errors.append('planes mode differs from the table plan')
    if gp.bytes_per_signal_launch != b * 4 + b * n * 8:
        errors.append('planes mode byte count')
    try:
        gp.launch(reward)                                # planes missing
        errors.append('planes mode without planes did not raise')
    except ValueError:
        pass
    try:
        StepGatherer(b, n, dev, reward_every=2)          # mode 'table': not allowed
        errors.append('reward_every with mode table did not raise')
    except ValueError:
        pass
    (Path(out_dir) / f'rank{rank}.txt').write_text('; '.join(errors) or 'ok')
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize('world', [2, 8])
def test_rewards_only_and_every_kth_step_gather_modes(tmp_path, world):
    mp.spawn(_modes_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    for rank in range(world):
        assert (tmp_path / f'rank{rank}.txt').read_text() == 'ok', rank


def _uneven_worker(rank, world, port, out_dir):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from gym_d2d_amd.distributed import StepGatherer, shard_range
    begin, end = shard_range(7, world, rank)                     # 4 + 3 envs
    try:
        StepGatherer(end - begin, 5, torch.device('cpu'))
        msg = 'no error'
    except ValueError as exc:
        msg = str(exc)
    (Path(out_dir) / f'rank{rank}.txt').write_text(msg)
    dist.barrier()
    dist.destroy_process_group()


def test_uneven_shards_are_rejected_up_front(tmp_path):
    """all_gather_into_tensor needs equal shards: StepGatherer must say so instead of hanging in the collective."""
    mp.spawn(_uneven_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    for rank in range(2):
        assert 'equal shards' in (tmp_path / f'rank{rank}.txt').read_text()


def _run_bench(args, env_extra=None, timeout=240):
    import json
    import subprocess
    env = dict(os.environ)
    env.pop('RANK', None); env.pop('WORLD_SIZE', None); env.pop('LOCAL_RANK', None)
    env.update(env_extra or {})
    r = subprocess.run([sys.executable, str(ROOT / 'bench.py')] + args, env=env, capture_output=True, text=True,
                       timeout=timeout)
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    return r, ([json.loads(lines[0])] if len(lines) == 1 and r.returncode == 0 else lines)


def test_bench_self_launches_its_ranks():
    """`python bench.py --gpus 2` with no torchrun around it: the parent spawns both ranks, rank 0 prints exactly one
    JSON line, the collectives saw 2 ranks and the gathered rewards are the ones the ranks produced (stub handle on
    CPU / gloo - the per-GPU step itself is covered by the -m gpu tests)."""
    r, out = _run_bench(['--gpus', '2', '--stub-cpu', '--steps', '3', '--warmup', '1', '--envs', '4', '--workload', 'default'])
    assert r.returncode == 0, r.stderr[-2000:]
    assert len(out) == 1 and isinstance(out[0], dict), r.stdout
    line = out[0]
    assert line['n_gpus'] == 2 and line['rccl_ranks'] == 2 and line['allreduce_rank_count'] == 2.0
    assert line['allgather_envs'] == 8 and line['checksums_agree'] is True
    # stub rewards: env block of rank r holds (r * 4 + steps_done) in every env -> sum over 8 envs
    steps_done = 4
    assert line['allreduce_reward_checksum'] == 4 * (0 + steps_done) + 4 * (4 + steps_done)
    assert line['value'] > 0 and line['config']['envs_per_gpu'] == 4 and 'stub' in line


def test_bench_self_launches_eight_ranks():
    """BASELINE config 5's world size through the launcher and the gather plumbing (stub handle, gloo), with the
    rewards-only gather plan."""
    r, out = _run_bench(['--gpus', '8', '--stub-cpu', '--steps', '3', '--warmup', '1', '--envs', '2', '--workload', 'default',
                         '--gather', 'rewards', '--reward-every', '4'], timeout=600)      # 1 + 3 launches = one ring block
    assert r.returncode == 0, r.stderr[-2000:]
    line = out[0]
    assert line['n_gpus'] == 8 and line['rccl_ranks'] == 8 and line['allreduce_rank_count'] == 8.0
    assert line['allgather_envs'] == 16 and line['checksums_agree'] is True and len(line['per_rank']) == 8
    assert line['gather']['mode'] == 'rewards' and line['gather']['bytes_per_gpu_per_step'] == 2 * 4 and line['gather']['reward_every'] == 4
    assert abs(line['value_per_gpu'] * 8 - line['value']) < 1e-6 * line['value']


def test_bench_launcher_deadline_ends_a_job_with_a_stuck_rank():
    """One rank never arrives (stuck before / in a collective): --rank-timeout ends the job - the launcher terminates its own
    children by PID and exits 124 - in well under twice the deadline, instead of holding the node until the driver's limit."""
    import time
    t0 = time.monotonic()
    r, out = _run_bench(['--gpus', '2', '--stub-cpu', '--steps', '2', '--warmup', '1', '--envs', '4', '--workload', 'default',
                         '--rank-timeout', '20'], env_extra={'D2D_BENCH_TEST_HANG_RANK': '1'}, timeout=120)
    took = time.monotonic() - t0
    assert r.returncode == 124, (r.returncode, r.stderr[-1500:])
    assert 'still running after --rank-timeout' in r.stderr and took < 40.0, took
    assert not any(isinstance(o, dict) for o in out)             # no result line from a job that did not finish


def test_bench_n_gt_1_line_says_where_the_cpu_baseline_is():
    r, out = _run_bench(['--gpus', '2', '--stub-cpu', '--steps', '2', '--warmup', '1', '--envs', '4', '--workload', 'default'])
    assert r.returncode == 0, r.stderr[-2000:]
    assert out[0]['cpu_baseline'].startswith('N=1 line only')


def test_bench_launcher_propagates_a_failing_rank():
    r, out = _run_bench(['--gpus', '2', '--stub-cpu', '--steps', '2', '--warmup', '1', '--envs', '4', '--workload', 'default'],
                        env_extra={'D2D_BENCH_TEST_FAIL_RANK': '1'})
    assert r.returncode != 0
    assert 'rank 1 exited' in r.stderr


def test_bench_runs_as_a_rank_under_an_external_launcher():
    """The driver's form: RANK / WORLD_SIZE already set by torch.distributed.run -> bench.py is a rank, not a launcher."""
    import subprocess
    port = str(_free_port())
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE='2', MASTER_ADDR='127.0.0.1', MASTER_PORT=port)
        procs.append(subprocess.Popen([sys.executable, str(ROOT / 'bench.py'), '--gpus', '2', '--stub-cpu', '--steps', '2',
                                       '--warmup', '1', '--envs', '4', '--workload', 'default'], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=240) for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    import json
    assert json.loads(outs[0][0].strip())['rccl_ranks'] == 2 and outs[1][0].strip() == ''
