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


def test_two_rank_gather_matches_single_process(tmp_path):
    world, global_envs = 2, 12
    mp.spawn(_worker, args=(world, _free_port(), global_envs, str(tmp_path)), nprocs=world, join=True)
    reward, table, obs = _oracle_results(global_envs, 0, global_envs)
    assert np.array_equal(np.load(tmp_path / 'reward.npy'), reward[:, 0])
    assert np.array_equal(np.load(tmp_path / 'table.npy'), table)
    # the learner-side expansion of the gathered compact table == the expanded obs every rank holds locally
    assert np.array_equal(np.load(tmp_path / 'obs.npy'), obs)
