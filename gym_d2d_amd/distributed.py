"""Multi-GPU: one process per GPU, the env axis sharded contiguously, no exchange inside the step.

Environments share nothing (interference, reward and observation all reduce within one env - SURVEY.md 8(e)), so the
only communication is what a single learner wants to see at the end of a step: one all-gather of the per-env rewards
(B_local floats) and of the COMPACT observation table T[B_local, N, 6].  The expanded LinearObs tensor [B, N, 6N]
is never sent: a consumer re-expands T locally (25.8 GB/GPU over one ~153 GB/s xGMI link per ring hop would take
seconds per step).

T is further split by how often it changes: its four position columns are constant within an episode
(simulator.py:61-75 - only reset() moves devices), so they are gathered once per reset (`gather_positions`); the per-step
gather carries only the (sinr_dB, snr_dB) columns [B_local, N, 2] - a third of the bytes (16.8 MB instead of 50 MB
per GPU per step at 4096 x 512), which matters because a ring all-gather is bound by ONE xGMI link.

Against the compact-obs step (about 30 us at 4096 x 512) even those 16.8 MB are forty steps' worth of ring time: StepGatherer
then gathers rewards only (mode='rewards', 16 KB per GPU) and / or ships the (sinr, snr) columns on every K-th step
(signal_every=K).

mode='planes' is the same plan for a learner whose ranks run D2D_OBS_NONE (no table is written at all, 24 of the compact
step's 64 bytes per link): the per-step gather takes the (sinr_dB, snr_dB) result PLANES [B_local, N] themselves - two
contiguous sends, no strided slice out of a table - and the positions come once per reset from the library's link rows
(D2D_BUF_LINK_POS [B_local, N, 4]); `table()` assembles the same [B_global, N, 6] on the consumer, bit for bit.

The per-step gather runs on a side stream from a staging copy, so it overlaps the obs-expansion kernel and the next
step; only the small device-to-device staging copy is ordered against the next step's writes.
Backend "nccl" is RCCL on ROCm; "gloo" works for CPU tensors (used by the CPU tests).
"""
from __future__ import annotations

from typing import Tuple

import torch
import torch.distributed as dist


def shard_range(global_envs: int, world_size: int, rank: int) -> Tuple[int, int]:
    """Contiguous block [begin, end) of env indices owned by `rank`; remainders go to the low ranks."""
    if not 0 <= rank < world_size:
        raise ValueError('rank out of range')
    base, extra = divmod(global_envs, world_size)
    begin = rank * base + min(rank, extra)
    return begin, begin + base + (1 if rank < extra else 0)


class StepGatherer:
    """All-gather of per-step results across ranks with equal shard sizes.

        g = StepGatherer(b_local, n_links, device)
        g.gather_positions(table)              # after every reset: the 4 position columns ([B,N,6] table or [B,N,4] link rows)
        g.launch(reward, table)                # after every step: rewards (+ (sinr, snr)), asynchronous
        g.launch(reward, sinr=s, snr=n)        # mode 'planes': the two result planes instead of the table
        rewards, signal = g.wait()             # [B_global], [B_global, N, 2]
        table = g.table()                      # [B_global, N, 6] assembled on demand

    mode 'table' (default): every launch gathers the rewards and the per-step columns of T (16.8 MB per GPU at 4096 x 512:
    sized against the 3.7 ms LinearObs step).  mode 'rewards': rewards only - B_local floats, 16 KB per GPU - for the
    compact-obs step (32 us at 4096 x 512), which a 16.8 MB ring all-gather (>= 0.3 ms over one ~100 GB/s xGMI link per
    hop) would outlast ten times over; the learner then reads observations through `signal_every`.
    mode 'planes': as 'table', but the per-step payload is read from the step's own sinr_db / snr_db planes (launch(reward,
    sinr=..., snr=...)) - for ranks that run D2D_OBS_NONE; rewards may be the per-env vector [B_local] (D2D_REWARD_PER_ENV).
    signal_every = K > 1: the (sinr, snr) columns ride along on every K-th launch only (the first included); `wait()` keeps
    returning the last gathered signal and `signal_step` says which launch it belongs to.
    With reward_every > 1 call `flush()` at the end of a rollout whose length is not a multiple of K: it gathers the partly
    filled ring and returns how many of its rows are valid.
    reward_every = K > 1 (mode 'rewards'): the host cost of ONE gather launch (stream hand-over, staging copy, collective: about 35 us
    through torch.distributed, measured with one rank) is more than a whole compact-obs step (about 28 us at 4096 x 512), so a
    per-step gather makes the step loop host bound.  With K > 1 every launch only copies the step's rewards into slot
    (launch mod K) of a device-side ring - one small copy on the CURRENT stream, no stream hand-over - and every K-th launch
    gathers the whole ring: `wait()` then returns the last K steps' rewards as [K, B_global] (row j = launch `reward_step` + j).
    per_agent_reward: gather rewards as [B, N] (Shannon / CueSinrShannon) instead of the env's scalar (SystemCapacity
    broadcasts one value to every agent, reward_fn.py:44: column 0 is all of it).
    timing: record CUDA events around every gather on the side stream; `gather_ms()` = mean ms per launch.

    backend 'torch' (default): torch.distributed collectives ("nccl" = RCCL on ROCm, "gloo" on CPU).
    backend 'native': the library's own RCCL entry (d2d_comm_init / d2d_allgather, include/d2d_hip.h) on `handle`;
    torch.distributed is then only used once, to ship rank 0's 128-byte unique id to the other ranks.  Every native
    collective - the per-episode position gather included - is issued on the ONE side stream, so two collectives of the
    communicator are never in flight on unordered streams.
    """

    def __init__(self, b_local: int, n_links: int, device: torch.device, group=None, *, backend: str = 'torch',
                 handle=None, mode: str = 'table', signal_every: int = 1, per_agent_reward: bool = False,
                 timing: bool = False, reward_every: int = 1) -> None:
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.device = device
        self.cuda = device.type == 'cuda'
        # all_gather_into_tensor needs the same shard size on every rank (shard_range may hand out uneven shards):
        # fail here with a clear message instead of hanging / corrupting inside the collective
        sizes = torch.tensor([b_local, -b_local], dtype=torch.int64, device=device)
        dist.all_reduce(sizes, op=dist.ReduceOp.MIN, group=group)
        lo, hi = int(sizes[0].item()), -int(sizes[1].item())
        if lo != hi:
            raise ValueError(f'StepGatherer needs equal shards on every rank, got between {lo} and {hi} envs per rank; '
                             'pad the batch to a multiple of the world size')
        if backend not in ('torch', 'native'):
            raise ValueError("backend must be 'torch' or 'native'")
        if mode not in ('table', 'rewards', 'planes'):
            raise ValueError("mode must be 'table', 'rewards' or 'planes'")
        if signal_every < 1:
            raise ValueError('signal_every must be >= 1')
        if reward_every < 1 or (reward_every > 1 and (mode != 'rewards' or per_agent_reward)):
            raise ValueError("reward_every > 1 needs mode='rewards' with per-env rewards")
        self.reward_every = int(reward_every)
        self.reward_step = -1                   # first launch of the gathered block (reward_every > 1)
        self.backend = backend
        self.handle = handle
        self.mode = mode
        self.signal_every = int(signal_every)
        self.per_agent_reward = bool(per_agent_reward)
        if backend == 'native':
            if handle is None or not self.cuda:
                raise ValueError("backend='native' needs the env's native handle and a CUDA device")
            box = [handle.comm_unique_id() if self.rank == 0 else None]
            dist.broadcast_object_list(box, src=0, group=group)
            handle.comm_init(self.world, self.rank, box[0])
        f32 = torch.float32
        rshape = (b_local, n_links) if per_agent_reward else (b_local,)
        if self.reward_every > 1:
            # two rings: launches fill one while the other is being gathered
            self.rings = [torch.zeros((self.reward_every, b_local), dtype=f32, device=device) for _ in range(2)]
            self.all_ring = torch.zeros((self.world, self.reward_every, b_local), dtype=f32, device=device)
            self._ring_free = [None, None]      # event: the gather that read this ring has finished
        self.stage_reward = torch.empty(rshape, dtype=f32, device=device)
        self.all_reward = torch.empty((self.world * b_local,) + rshape[1:], dtype=f32, device=device)
        self.stage_signal = self.all_signal = None
        if mode == 'table':
            self.stage_signal = torch.empty((b_local, n_links, 2), dtype=f32, device=device)
            self.all_signal = torch.zeros((self.world * b_local, n_links, 2), dtype=f32, device=device)
        self.stage_planes = self.all_planes = None
        if mode == 'planes':                    # [0] sinr_dB, [1] snr_dB: each plane contiguous on both sides of the collective
            self.stage_planes = [torch.empty((b_local, n_links), dtype=f32, device=device) for _ in range(2)]
            self.all_planes = [torch.zeros((self.world * b_local, n_links), dtype=f32, device=device) for _ in range(2)]
        self.all_positions = torch.zeros((self.world * b_local, n_links, 4), dtype=f32, device=device)
        self.launches = 0
        self._ring_pos = 0                      # ring slots consumed (launches + the slots a flush() skipped)
        self.signal_step = -1                   # index of the launch the gathered signal belongs to
        self.bytes_per_launch = self.stage_reward.numel() * 4       # per step (reward_every > 1: K of them travel together)
        self.bytes_per_signal_launch = self.bytes_per_launch + (b_local * n_links * 8 if mode in ('table', 'planes') else 0)
        self._timing = bool(timing) and self.cuda
        self._events = []
        self._launch_mark = 0
        if self.cuda:
            self.comm_stream = torch.cuda.Stream(device=device)
            self.staged = torch.cuda.Event()
            self.done = torch.cuda.Event()
        self._pending = False

    def _all_gather(self, out: torch.Tensor, local: torch.Tensor, stream_ptr: int = 0) -> None:
        if self.backend == 'native':
            self.handle.allgather(local.data_ptr(), out.data_ptr(), local.numel() * local.element_size(), stream_ptr)
        else:
            dist.all_gather_into_tensor(out, local, group=self.group)

    def gather_positions(self, table: torch.Tensor) -> torch.Tensor:
        """Once per episode: all ranks' (tx_x, tx_y, rx_x, rx_y) columns of T -> [B_global, N, 4].  `table` is the obs table
        [B_local, N, 6] or the link-position rows [B_local, N, 4] themselves (D2D_BUF_LINK_POS: mode 'planes', where no table
        exists).  Ordered after the work already enqueued on the current stream and before whatever the caller enqueues next."""
        if table.shape[-1] not in (4, 6):
            raise ValueError('gather_positions takes the [B, N, 6] obs table or the [B, N, 4] link-position rows')
        if self.cuda:
            cur = torch.cuda.current_stream(self.device)
            self.comm_stream.wait_stream(cur)                   # the reset's rows are ready; earlier gathers precede us there
            with torch.cuda.stream(self.comm_stream):
                local = table[:, :, :4].contiguous()
                local.record_stream(self.comm_stream)
                self._all_gather(self.all_positions, local, self.comm_stream.cuda_stream)
            cur.wait_stream(self.comm_stream)
        else:
            self._all_gather(self.all_positions, table[:, :, :4].contiguous())
        return self.all_positions

    def launch(self, reward_per_agent: torch.Tensor, table: torch.Tensor = None, *, sinr: torch.Tensor = None,
               snr: torch.Tensor = None) -> None:
        """Call right after the step was enqueued on the current stream.  reward_per_agent [B_local, N] (column 0
        is the env's scalar for SystemCapacity) or the per-env vector [B_local] (D2D_REWARD_PER_ENV); table [B_local, N, 6]
        (mode 'table'); sinr / snr [B_local, N] (mode 'planes')."""
        if self.reward_every > 1:
            return self._launch_ring(reward_per_agent)
        with_signal = self.mode in ('table', 'planes') and self.launches % self.signal_every == 0
        if with_signal and self.mode == 'table' and table is None:
            raise ValueError("mode 'table' gathers the (sinr, snr) columns: pass the obs table")
        planes = self.mode == 'planes'
        if with_signal and planes and (sinr is None or snr is None):
            raise ValueError("mode 'planes' gathers the sinr_db / snr_db planes: pass sinr= and snr=")
        if reward_per_agent.dim() == 1:
            if self.per_agent_reward:
                raise ValueError('per_agent_reward needs the [B, N] reward rows')
            src_reward = reward_per_agent
        else:
            src_reward = reward_per_agent if self.per_agent_reward else reward_per_agent[:, 0]
        if self.cuda:
            cur = torch.cuda.current_stream(self.device)
            self.comm_stream.wait_stream(cur)                   # results of this step are ready
            with torch.cuda.stream(self.comm_stream):
                if self._timing:
                    e0 = torch.cuda.Event(enable_timing=True); e0.record(self.comm_stream)
                self.stage_reward.copy_(src_reward)
                if with_signal and planes:
                    self.stage_planes[0].copy_(sinr); self.stage_planes[1].copy_(snr)      # contiguous: plain block copies
                elif with_signal:
                    self.stage_signal.copy_(table[:, :, 4:6])
                self.staged.record(self.comm_stream)
                self._all_gather(self.all_reward, self.stage_reward, self.comm_stream.cuda_stream)
                if with_signal and planes:
                    for k in range(2):
                        self._all_gather(self.all_planes[k], self.stage_planes[k], self.comm_stream.cuda_stream)
                elif with_signal:
                    self._all_gather(self.all_signal, self.stage_signal, self.comm_stream.cuda_stream)
                self.done.record(self.comm_stream)
                if self._timing:
                    e1 = torch.cuda.Event(enable_timing=True); e1.record(self.comm_stream)
                    self._record(e0, e1)
            cur.wait_event(self.staged)                         # next step may overwrite table/reward now
        else:
            self.stage_reward.copy_(src_reward)
            self._all_gather(self.all_reward, self.stage_reward)
            if with_signal and planes:
                for k, src in enumerate((sinr, snr)):
                    self.stage_planes[k].copy_(src)
                    self._all_gather(self.all_planes[k], self.stage_planes[k])
            elif with_signal:
                self.stage_signal.copy_(table[:, :, 4:6])
                self._all_gather(self.all_signal, self.stage_signal)
        if with_signal:
            self.signal_step = self.launches
        self.launches += 1
        self._pending = True

    def _launch_ring(self, reward_per_agent: torch.Tensor) -> None:
        k = self.reward_every
        slot, which = self._ring_pos % k, (self._ring_pos // k) % 2
        ring = self.rings[which]
        src = reward_per_agent if reward_per_agent.dim() == 1 else reward_per_agent[:, 0]
        if self.cuda:
            cur = torch.cuda.current_stream(self.device)
            if slot == 0 and self._ring_free[which] is not None:
                cur.wait_event(self._ring_free[which])          # the gather that read this ring two blocks ago is done (long since)
            ring[slot].copy_(src)                               # on the current stream: ordered with the step, no hand-over
        else:
            ring[slot].copy_(src)
        if slot == k - 1:
            self._gather_ring(which, self.launches - (k - 1))
        self.launches += 1
        self._ring_pos += 1

    def _gather_ring(self, which: int, first_step: int) -> None:
        ring = self.rings[which]
        if self.cuda:
            cur = torch.cuda.current_stream(self.device)
            self.comm_stream.wait_stream(cur)
            with torch.cuda.stream(self.comm_stream):
                if self._timing:
                    e0 = torch.cuda.Event(enable_timing=True); e0.record(self.comm_stream)
                self._all_gather(self.all_ring.view(-1), ring.view(-1), self.comm_stream.cuda_stream)
                self.done.record(self.comm_stream)
                free = torch.cuda.Event(); free.record(self.comm_stream)
                self._ring_free[which] = free
                if self._timing:
                    e1 = torch.cuda.Event(enable_timing=True); e1.record(self.comm_stream)
                    self._record(e0, e1)
        else:
            self._all_gather(self.all_ring.view(-1), ring.view(-1))
        self.reward_step = first_step
        self.reward_valid = self.reward_every                   # a whole block (flush() lowers it for a partial one)
        self._pending = True

    def flush(self) -> int:
        """reward_every > 1: gather the ring as it stands when the rollout did not end on a block boundary (launches % K != 0).
        Returns how many rows are valid (launches `reward_step` ... `reward_step` + valid - 1); `wait()` then returns exactly
        those rows, [valid, B_global] (`reward_valid` holds the count); 0 = nothing was pending.  A COLLECTIVE: every rank must
        call it at the same launch count - ranks that disagree on launches % K would leave some inside the all-gather and
        others outside it until the rank timeout.  The next launch starts a new block."""
        k = self.reward_every
        valid = self._ring_pos % k if k > 1 else 0
        if valid == 0:
            return 0
        self._gather_ring((self._ring_pos // k) % 2, self.launches - valid)
        self.reward_valid = valid
        self._ring_pos += k - valid                             # the next launch opens the next block (slot 0, the other ring)
        return valid

    def _record(self, e0, e1) -> None:
        self._events.append((e0, e1))
        if len(self._events) > 8192:                            # long timed runs: fold the older half into a running sum
            drop = len(self._events) // 2
            self._events[drop - 1][1].synchronize()
            self._dropped_ms = getattr(self, '_dropped_ms', 0.0) + sum(a.elapsed_time(b) for a, b in self._events[:drop])
            del self._events[:drop]

    def wait(self) -> Tuple[torch.Tensor, torch.Tensor]:
        """(rewards [B_global] (or [B_global, N]), signal [B_global, N, 2] = sinr_dB, snr_dB - None in mode 'rewards'; in
        mode 'planes' the pair of planes (sinr_dB, snr_dB), each [B_global, N]) of
        the last launched gather, rank-major = global env order.  With signal_every > 1 the signal is that of launch
        `signal_step`."""
        if self.cuda and self._pending:
            torch.cuda.current_stream(self.device).wait_event(self.done)
        self._pending = False
        if self.reward_every > 1:
            # [world, K, b_local] -> [K, B_global]: rank-major along the env axis = global env order
            # (after a flush(): only the rows the partial block filled - the rest of the ring holds an older block)
            return self.all_ring.permute(1, 0, 2).reshape(self.reward_every, -1)[:getattr(self, 'reward_valid', self.reward_every)], None
        if self.mode == 'planes':
            return self.all_reward, tuple(self.all_planes)      # (sinr_dB [B_global, N], snr_dB [B_global, N])
        return self.all_reward, self.all_signal

    def table(self) -> torch.Tensor:
        """The global compact obs table [B_global, N, 6] = cached positions ++ latest gathered (sinr, snr)."""
        _, signal = self.wait()
        if signal is None:
            raise ValueError("mode 'rewards' gathers no observation columns")
        if self.mode == 'planes':
            return torch.cat([self.all_positions, signal[0].unsqueeze(2), signal[1].unsqueeze(2)], dim=2)
        return torch.cat([self.all_positions, signal], dim=2)

    def gather_ms(self) -> float:
        """Mean milliseconds per launch between the gather's first staging copy and its last collective, measured with
        events on the side stream (timing=True); synchronises.  0.0 when nothing was timed."""
        if not self._events:
            return 0.0
        self._events[-1][1].synchronize()
        launches = max(1, self.launches - self._launch_mark)      # reward_every > 1: one gather per K launches
        return (getattr(self, '_dropped_ms', 0.0) + sum(e0.elapsed_time(e1) for e0, e1 in self._events)) / launches

    def reset_timing(self) -> None:
        if self._events:
            self._events[-1][1].synchronize()
        self._events = []
        self._dropped_ms = 0.0
        self._launch_mark = self.launches


def expand_table(table: torch.Tensor, handle=None, out: torch.Tensor = None) -> torch.Tensor:
    """Consumer-side LinearObs expansion of a gathered table [B, N, 6] -> [B, N, 6N] (what a learner on another GPU
    does instead of receiving the expanded tensor).  obs[b,i] = (T[i], T[0..i-1], T[i+1..]) (obs_fn.py:43-53).

    CUDA tensors run the library's own expansion kernel through d2d_expand_table on `handle` (any native handle on
    that GPU; bit-identical to the D2D_BUF_OBS the owning rank produced) on torch's current stream.  CPU tensors -
    the gloo tests - use torch indexing."""
    b, n, w = table.shape
    if table.is_cuda:
        if handle is None:
            raise ValueError('expand_table on a CUDA tensor needs a native handle (d2d_expand_table); there is no '
                             'generic-torch fallback on the GPU')
        table = table.contiguous()
        if out is None:
            out = torch.empty((b, n, n * w), dtype=torch.float32, device=table.device)
        handle.set_stream(torch.cuda.current_stream(table.device).cuda_stream)
        handle.expand_table(table.data_ptr(), b, n, out.data_ptr())
        return out
    return expand_table_torch(table)


def expand_table_torch(table: torch.Tensor) -> torch.Tensor:
    """The same expansion with torch advanced indexing (CPU tensors; reference for the kernel in the tests)."""
    b, n, w = table.shape
    idx = torch.arange(n, device=table.device)
    k = torch.arange(n, device=table.device)
    # source link for slot k of agent i: i for k = 0; k-1 for 1 <= k <= i; k for k > i
    src = torch.where(k[None, :] == 0, idx[:, None], torch.where(k[None, :] <= idx[:, None], k[None, :] - 1, k[None, :]))
    return table[:, src, :].reshape(b, n, n * w)
