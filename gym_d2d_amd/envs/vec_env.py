"""VecD2DEnv - the batched, array-native face of D2DEnv: B independent environments stepped by one kernel launch.

The reference has no batched API (dicts of ~2M Python objects per step are not viable); this class keeps D2DEnv's
semantics per env (reset = new positions + one step with random CUE/DUE actions, 10-step episodes, same decode,
same reward / obs definitions) but speaks arrays:

    env = VecD2DEnv({'num_rbs': 256, 'num_cues': 256, 'num_due_pairs': 256}, num_envs=4096)
    obs = env.reset(seed=1234)                        # [B, N, 6N] float32 (LinearObsFunction)
    obs, rewards, dones, info = env.step(actions)     # actions int32 [B, N]; rewards [B, N]; dones [B] bool

With PyTorch-ROCm present all arrays are CUDA tensors that alias the library's HBM buffers (zero copy: the tensors are
allocated by torch and bound into the handle, kernels run on torch's current stream).  Without torch they are NumPy
copies.  Agent order along N: all CUE uplinks, then all DUE sidelinks (d2d_env.py:54-60).
"""
from __future__ import annotations

from types import SimpleNamespace
from typing import Optional

import numpy as np

from .. import _native
from ..simulator import BASE_STATION_ID, Simulator
from ..traffic_model import DownlinkTrafficModel
from . import _rng
from .d2d_env import EPISODE_LENGTH
from .obs_fn import ArrayObsFunction, LinearObsFunction, ObsFunction
from .reward_fn import SystemCapacityRewardFunction

try:
    import torch
except Exception:       # pragma: no cover
    torch = None

_OUTPUTS = (('sinr_db', _native.BUF_SINR_DB), ('snr_db', _native.BUF_SNR_DB), ('rate_bps', _native.BUF_RATE_BPS),
            ('capacity_mbps', _native.BUF_CAPACITY), ('reward', _native.BUF_REWARD))


class VecD2DEnv:
    def __init__(self, env_config: Optional[dict] = None, num_envs: Optional[int] = None, *,
                 cue_actions: str = 'agent', use_torch: Optional[bool] = None, first_env: int = 0,
                 export_actions: bool = True, reward_per_env: bool = False, placement_trials: int = 0,
                 placement_budget_bytes: int = 1 << 30) -> None:
        """cue_actions: 'agent' - step() takes actions for CUEs and DUEs [B, C+P] (reference behaviour);
        'traffic' - CUE links follow the env's traffic model (round-robin RB at max power,
        traffic_model.py:15-22): their (rb, pwr) are constants of the kernel's link records
        (d2d_set_fixed_actions) and step() takes DUE actions only [B, P] - the CUE half of the action
        traffic does not exist.
        first_env: global index of env 0 when one logical batch is sharded over several GPUs (positions AND the
        random actions of reset() are keyed by global env index, so a sharded run reproduces the single-GPU run).

        export_actions: False - the kernel does not write the decoded (rb, tx power) planes behind info['rb'] /
        info['tx_pwr_dbm'] (d2d_set_export_actions: 8 of the step's 72 bytes per link; a rollout knows its own actions);
        the two info entries are then None.

        reward_per_env: True - SystemCapacityRewardFunction's reward is ONE scalar per env handed to every agent
        (reward_fn.py:42-44); step() then returns it as [B] instead of N copies [B, N] (d2d_set_reward_layout: 4 bytes per
        link and step less).  Only with the native SystemCapacity reward.

        placement_trials (default 0 = off; opt-in): where the step's dominant output block sits physically decides how fast the
        kernel streams it - the same 61 MB obs block of BASELINE config 2 runs 13.2, 13.9 or 15.1 us per step depending on the
        allocation it landed in, reproducibly per allocation (profiles/r4_obs_block_placement_candidates.jsonl; TLB and
        L2-channel counters are equal, so it is how the block's pages spread over the memory channels).  K > 1: the first
        reset() times a few hundred steps on up to K candidate blocks for the fused LinearObs step's obs block (or the table of a
        compact-obs env), allocated one by one behind paddings of 2 - 22 MB, and keeps the fastest; it stops early once a
        candidate is 7 % faster than the slowest seen.  Bounded and private: candidates + paddings never exceed
        placement_budget_bytes (default 1 GiB) at once, each comes from a torch MemPool of its own that is dropped with the
        loser (nothing is released from - or left in - the caller's caching allocator: no torch.cuda.empty_cache()), and the
        obs tensor step() returns MOVES once, at that first reset (a tensor taken from an earlier reset() of the same env does
        not exist - the trials run inside the first).  A lottery with better odds, not a cure: off unless asked for.

        step()'s `dones` on the torch path is one of two preallocated CONSTANT tensors (all False / all True), shared by
        every call: treat it as read-only (clone it before an in-place update).

        env_config['obs_dtype'] = 'float64' returns observations in the reference's dtype (obs_fn.py:51 builds float64
        arrays); the default float32 is the kernels' own block, zero copy.

        reset() checks the status flags once and raises ValueError('math domain error') if two interacting devices
        coincide (what the reference's log10(0) does, path_loss.py:66); step() does not synchronise - poll
        status_flags() for FLAG_ZERO_DISTANCE / FLAG_NON_FINITE if positions are written from outside."""
        env_config = dict(env_config or {})
        self.export_actions = bool(export_actions)
        obs_cls = env_config.pop('obs_fn', LinearObsFunction)
        rew_cls = env_config.pop('reward_fn', SystemCapacityRewardFunction)
        if num_envs is not None:
            env_config['num_envs'] = num_envs
        self.obs_fn = obs_cls()
        self.reward_fn = rew_cls()
        if isinstance(self.obs_fn, ObsFunction) and type(self.obs_fn).get_state is not LinearObsFunction.get_state:
            raise TypeError('dict-style ObsFunction subclasses work with D2DEnv; VecD2DEnv takes LinearObsFunction or '
                            'an ArrayObsFunction')
        rid = getattr(self.reward_fn, 'native_id', _native.REWARD_NONE)
        if rid == _native.REWARD_NONE and not hasattr(self.reward_fn, 'compute'):
            raise TypeError('VecD2DEnv takes a built-in reward function or an object with compute(view) -> [B,N]')

        n_links = int(env_config.get('num_cues', 25)) + int(env_config.get('num_due_pairs', 25))
        self.simulator = Simulator(env_config, max_links=n_links)     # link set is fixed: no spare capacity needed
        sim, cfg = self.simulator, self.simulator.config
        self.num_envs = sim.num_envs
        self.config = cfg
        self.observation_space = self.obs_fn.get_obs_space(cfg)
        self.num_pwr_actions = cfg.num_pwr_actions
        # CUE links: uplinks cueXX -> mbs (what reset() builds, d2d_env.py:54-60), unless the CUEs are driven by a
        # DownlinkTrafficModel, whose links are mbs -> cueXX (traffic_model.py:25-32)
        self._cue_kind = 'cue'
        if cue_actions == 'traffic' and isinstance(sim.traffic_model, DownlinkTrafficModel):
            self._cue_kind = 'mbs'
            sim.set_links([(BASE_STATION_ID, c) for c in sim.devices.cues.keys()] + list(sim.devices.dues.keys()))
        else:
            sim.set_links(sim.default_link_keys())
        self.num_cues, self.num_due_pairs = cfg.num_cues, cfg.num_due_pairs
        self.num_links = self.num_cues + self.num_due_pairs
        if cue_actions not in ('agent', 'traffic'):
            raise ValueError("cue_actions must be 'agent' or 'traffic'")
        self.cue_actions = cue_actions
        self.num_agents = self.num_links if cue_actions == 'agent' else self.num_due_pairs

        h = sim.handle
        self.first_env = int(first_env)
        if cue_actions == 'traffic' and self.num_cues:
            # (rb, pwr) exactly as TrafficModel.get_traffic would hand them to the simulator: no a // P encode /
            # decode round trip, so a device_config_file max_tx_power_dBm above the action alphabet stays legal
            rb, pwr = sim.traffic_model.assignments(sim.devices)
            h.set_fixed_actions(np.arange(self.num_cues), rb, pwr)
        h.set_env_offset(first_env)
        h.set_export_actions(self.export_actions)
        self._obs64 = cfg.obs_dtype == 'float64'
        h.set_obs_mode(self.obs_fn.native_mode)
        # float64 observations of the native LinearObs: the expansion kernel writes them as float64 itself (no cast pass)
        self._native_obs64 = self._obs64 and self.obs_fn.native_mode == _native.OBS_LINEAR and not isinstance(self.obs_fn, ArrayObsFunction)
        if self._native_obs64:
            h.set_obs_dtype(_native.F64)
        h.set_reward(rid, float(getattr(self.reward_fn, 'native_param', 0.0)))
        self._native_reward = bool(rid)
        self.reward_per_env = bool(reward_per_env)
        if self.reward_per_env:
            if rid != _native.REWARD_SYSTEM_CAPACITY:
                raise ValueError('reward_per_env needs the native SystemCapacityRewardFunction (the per-agent rewards differ per link)')
            h.set_reward_layout(_native.REWARD_PER_ENV)
        self._array_obs = isinstance(self.obs_fn, ArrayObsFunction)
        self.use_torch = (torch is not None and torch.cuda.is_available()) if use_torch is None else use_torch
        self._t = {}
        if self.use_torch:
            self._bind_torch_buffers()
        self.num_steps = 0
        self._episode = 0
        self._seed = cfg.seed if cfg.seed is not None else 0
        # which block the trials move: the obs block of the fused LinearObs step, or the table of the compact-obs step
        mode = self.obs_fn.native_mode
        fused = mode == _native.OBS_LINEAR and self.num_links <= 128 and not getattr(self, '_native_obs64', False)
        self._placement_target = ('obs', _native.BUF_OBS) if fused else (('table', _native.BUF_OBS_TABLE) if mode == _native.OBS_TABLE else None)
        if placement_trials in ('auto', None):             # the default of rounds 3 - 4; since round 5 the trials are opt-in
            placement_trials = 0
        if not isinstance(placement_trials, int) or isinstance(placement_trials, bool) or placement_trials < 0:
            raise ValueError("placement_trials must be a non-negative int (0 = off; 'auto' is accepted as 0)")
        self._placement_trials = placement_trials if self.use_torch and self._placement_target is not None else 0
        self._placement_budget = int(placement_budget_bytes)
        self.placement = None                      # after the trials: {'buffer': ..., 'us_per_step': [...], 'chosen': k}

    # ------------------------------------------------------------------ buffers
    def _bind_torch_buffers(self) -> None:
        h = self.simulator.handle
        dev = torch.device('cuda', self.config.device_ordinal)
        self.device = dev
        b, n, d = self.num_envs, self.num_links, h.num_devices
        cap = h.max_links

        def alloc(which, shape, dtype):
            # capacity follows the handle's max_links so re-linking never outgrows the binding
            t = torch.empty(shape, dtype=dtype, device=dev)
            h.bind_buffer(which, t.data_ptr(), t.numel() * t.element_size())
            return t

        self._t['pos_x'] = alloc(_native.BUF_POS_X, (b, d), torch.float32)
        self._t['pos_y'] = alloc(_native.BUF_POS_Y, (b, d), torch.float32)
        self._t['actions'] = alloc(_native.BUF_ACTIONS, (b * cap,), torch.int32)[:b * self.num_agents].view(b, self.num_agents)
        for name, which in (('rb', _native.BUF_RB), ('pwr', _native.BUF_PWR)):
            self._t[name] = alloc(which, (b * cap,), torch.int32)[:b * n].view(b, n)
        for name, which in _OUTPUTS:
            if name == 'reward' and self.reward_per_env:
                self._t[name] = alloc(_native.BUF_REWARD_ENV, (b,), torch.float32)
                continue
            self._t[name] = alloc(which, (b * cap,), torch.float32)[:b * n].view(b, n)
        if self.obs_fn.native_mode != _native.OBS_NONE:
            self._t['table'] = alloc(_native.BUF_OBS_TABLE, (b * cap * 6,), torch.float32)[:b * n * 6].view(b, n, 6)
        else:
            self._t['table'] = None                        # D2D_OBS_NONE: no table exists; positions: link_positions()
        self._t['env_flags'] = alloc(_native.BUF_ENV_FLAGS, (b,), torch.int32)
        if self.obs_fn.native_mode == _native.OBS_LINEAR:
            self._t['obs'] = alloc(_native.BUF_OBS, (b, n, 6 * n), torch.float64 if self._native_obs64 else torch.float32)
        self._stream_ptr = None
        # the raw current-stream query (one C call, no Stream object) when this torch has it
        raw = getattr(torch._C, '_cuda_getCurrentRawStream', None)
        index = dev.index if dev.index is not None else torch.cuda.current_device()
        self._current_stream_ptr = (lambda: raw(index)) if raw is not None else \
            (lambda: torch.cuda.current_stream(self.device).cuda_stream)
        self._follow_torch_stream()
        # per-step constants of the torch path: the buffers do not move (the opt-in placement trials re-bind ONE block once, inside
        # the first reset, and drop this cache), so the view, the info dict and the two possible `dones` vectors are built once
        self._view_cache = None
        self._dones = (torch.zeros(b, dtype=torch.bool, device=dev), torch.ones(b, dtype=torch.bool, device=dev))

    def _follow_torch_stream(self) -> None:
        """Run the library's kernels on torch's CURRENT stream (the null stream by default) so they are ordered with
        the torch ops that produce actions / consume results.  Re-bound only when the caller switches streams."""
        ptr = self._current_stream_ptr()
        if ptr != self._stream_ptr:
            self.simulator.handle.set_stream(ptr)
            self._stream_ptr = ptr

    def _view(self) -> SimpleNamespace:
        sim = self.simulator
        if self.use_torch:
            if self._view_cache is not None:
                return self._view_cache
            v = dict(self._t)
        else:
            v = {name: sim.fetch(_native.BUF_REWARD_ENV if name == 'reward' and self.reward_per_env else which)
                 for name, which in _OUTPUTS if name != 'reward' or getattr(self.reward_fn, 'native_id', 0)}
            if self.export_actions:                        # otherwise the two planes are stale / never written
                v['rb'] = sim.fetch(_native.BUF_RB); v['pwr'] = sim.fetch(_native.BUF_PWR)
            v['table'] = sim.fetch(_native.BUF_OBS_TABLE) if self.obs_fn.native_mode != _native.OBS_NONE else None
            v['pos_x'] = sim.fetch(_native.BUF_POS_X); v['pos_y'] = sim.fetch(_native.BUF_POS_Y)
            if self.obs_fn.native_mode == _native.OBS_LINEAR:
                v['obs'] = sim.fetch(_native.BUF_OBS)
        v.update(link_tx=sim.link_tx, link_rx=sim.link_rx, link_type=sim.link_type)
        if not self.export_actions:
            v['rb'] = v['pwr'] = None
        view = SimpleNamespace(**v)
        if self.use_torch:
            self._view_cache = view
            self._info = {'rb': view.rb, 'tx_pwr_dbm': view.pwr, 'snr_db': view.snr_db, 'sinr_db': view.sinr_db,
                          'rate_bps': view.rate_bps, 'capacity_mbps': view.capacity_mbps}
        return view

    # ------------------------------------------------------------------ gym-like API
    def _initial_action_highs(self):
        n_cue = self.config.num_rbs * self.num_pwr_actions[self._cue_kind]
        n_due = self.config.num_rbs * self.num_pwr_actions['due']
        return ([n_cue] * self.num_cues if self.cue_actions == 'agent' else []) + [n_due] * self.num_due_pairs

    def reset(self, seed: Optional[int] = None):
        """New positions for every env (device-side sampler), then one step with uniformly random actions on every
        agent-driven link to produce the initial SINRs (d2d_env.py:45-60).  Both draws are counter-based and keyed
        by global env index (first_env + b)."""
        if seed is not None:
            self._seed, self._episode = int(seed), 0
        self.num_steps = 0
        if self.use_torch:
            self._follow_torch_stream()
        self.simulator.reset_device(self._seed, self._episode)
        highs = self._initial_action_highs()
        if self.use_torch:
            if self.num_agents:
                self._t['actions'].copy_(_rng.uniform_ints_torch(torch, self._seed, self._episode, self.first_env,
                                                                  self.num_envs, self.num_agents, highs, self.device))
            self.simulator.handle.step()
        else:
            a = _rng.uniform_ints_numpy(self._seed, self._episode, self.first_env, self.num_envs, self.num_agents, highs) \
                if self.num_agents else np.zeros((self.num_envs, 0), np.int32)
            self.simulator.step_arrays(a)
        self._episode += 1
        if self.simulator.handle.status_flags() & _native.FLAG_ZERO_DISTANCE:
            raise ValueError('math domain error')            # log10(0) in path_loss.py:66
        if self._placement_trials > 1 and self.placement is None:
            self._choose_obs_placement(self._placement_trials)
        return self._observe(self._view())

    def _choose_obs_placement(self, trials: int, warm_ms: float = 15.0, steps: int = 256) -> None:
        """Time the step on up to `trials` candidate blocks for the dominant output (all held until the choice, hence distinct
        physical ranges), keep the fastest.  The step repeated here is the reset's own (same positions, same actions in the
        bound action buffer: same outputs), so the env's state after the trials is what reset() produced; only the shadowing
        model's step counter would advance, so that model keeps the block it has."""
        import time
        h = self.simulator.handle
        if getattr(self.simulator, 'shadowing_seed', None) is not None:
            self.placement = {'skipped': 'ShadowingPathLoss draws per step'}
            return
        key, which = self._placement_target
        base = self._t[key]                                     # a view of the bound allocation (capacity may exceed the active part)
        first = base._base if base._base is not None else base
        nbytes = first.numel() * first.element_size()

        def timed(n):
            torch.cuda.synchronize(self.device)
            t0 = time.perf_counter()
            for _ in range(n):
                h.step()
            torch.cuda.synchronize(self.device)
            return (time.perf_counter() - t0) / n * 1e6
        t_end = time.perf_counter() + warm_ms * 1e-3            # past the clock ramp behind the idle stretch of building the env
        while time.perf_counter() < t_end:
            timed(64)
        # every candidate (and the padding in front of it) lives in a MemPool of its own: dropping a loser's pool gives its memory
        # back to the driver without touching the caching allocator the caller's tensors live in
        def alloc_private(nbytes_, like=None):
            pool = torch.cuda.MemPool() if hasattr(torch.cuda, 'MemPool') else None
            if pool is None:
                return None, (torch.empty_like(like) if like is not None else torch.empty(nbytes_, dtype=torch.uint8, device=self.device))
            # use_mem_pool routes the allocations of ONE device: name the env's, which need not be the current one
            with torch.cuda.device(self.device), torch.cuda.use_mem_pool(pool, device=self.device):
                t = torch.empty_like(like) if like is not None else torch.empty(nbytes_, dtype=torch.uint8, device=self.device)
            return pool, t
        cands, pools, pads, times = [first], [None], [], []
        held = 0
        for k in range(trials):
            if k:
                # neighbouring allocations tend to share a speed class: a padding of varying size in front of every candidate
                pad = ((k * 7) % 11 + 1) * (2 << 20) + (k % 3) * 4096
                if held + pad + nbytes > self._placement_budget:
                    break
                try:
                    pads.append(alloc_private(pad))
                    pool, t = alloc_private(0, like=first)
                except torch.cuda.OutOfMemoryError:             # a crowded GPU: choose among what there is
                    break
                cands.append(t); pools.append(pool)
                held += pad + nbytes
            h.bind_buffer(which, cands[k].data_ptr(), nbytes)
            timed(32)
            times.append(timed(steps))
            if k >= 3 and min(times) < 0.93 * max(times):
                break
        best = min(range(len(times)), key=times.__getitem__)
        chosen = cands[best]
        h.bind_buffer(which, chosen.data_ptr(), nbytes)
        self._t[key] = chosen[:base.numel()].view(base.shape) if chosen.shape != base.shape else chosen
        self._placement_pool = pools[best]                      # the winner's pool lives as long as the env
        h.step()                                                # the reset's step once more, into the block that stays
        self._view_cache = None
        self.placement = {'buffer': key, 'us_per_step': [round(t, 2) for t in times], 'chosen': best,
                          'transient_bytes': held, 'private_pools': pools[-1] is not None or len(pools) == 1}
        # the losers and the paddings: tensors first, then their pools (a pool without live tensors returns its blocks to the driver)
        torch.cuda.synchronize(self.device)
        del cands, pads, first, base, chosen, pools

    def step(self, actions):
        """actions: int [B, num_agents] (torch CUDA tensor, or NumPy).  Returns (obs, rewards[B,N], dones[B], info).
        Asynchronous on the torch path: nothing here waits for the GPU, so error flags are NOT checked per step -
        see status_flags()."""
        sim = self.simulator
        if self.use_torch:
            self._follow_torch_stream()
            src = actions if torch.is_tensor(actions) else torch.as_tensor(np.asarray(actions), device=self.device)
            if tuple(src.shape) != (self.num_envs, self.num_agents):
                raise ValueError(f'actions must be [{self.num_envs},{self.num_agents}], got {tuple(src.shape)}')
            if src.dtype == torch.int32 and src.is_contiguous() and src.device == self.device:
                sim.handle.step(src.data_ptr())               # zero copy: the kernel reads the caller's tensor
            else:
                self._t['actions'].copy_(src)
                sim.handle.step()
        else:
            src = np.asarray(actions, dtype=np.int32)
            if src.shape != (self.num_envs, self.num_agents):
                raise ValueError(f'actions must be [{self.num_envs},{self.num_agents}], got {src.shape}')
            sim.step_arrays(src)
        self.num_steps += 1
        view = self._view()
        obs = self._observe(view)
        rewards = view.reward if self._native_reward else self.reward_fn.compute(view)
        done = self.num_steps >= EPISODE_LENGTH
        if self.use_torch:
            return obs, rewards, self._dones[done], dict(self._info)
        info = {'rb': view.rb, 'tx_pwr_dbm': view.pwr, 'snr_db': view.snr_db, 'sinr_db': view.sinr_db,
                'rate_bps': view.rate_bps, 'capacity_mbps': view.capacity_mbps}
        return obs, rewards, np.full(self.num_envs, done), info

    def _observe(self, view):
        obs = self.obs_fn.compute(view) if self._array_obs else view.obs
        if self._obs64 and not self._native_obs64 and not isinstance(obs, tuple):   # the reference's dtype (obs_fn.py:51) for a custom array obs: cast here
            obs = obs.double() if self.use_torch else np.asarray(obs, dtype=np.float64)
        return obs

    def link_positions(self):
        """[B, N, 4] float32 (tx_x, tx_y, rx_x, rx_y) of every link = columns 0-3 of the obs table (obs_fn.py:57-59), constant
        between resets.  torch path: a zero-copy view of the library's own rows (D2D_BUF_LINK_POS) - read-only, valid until
        close(), refreshed by the library on reset."""
        h = self.simulator.handle
        if not self.use_torch:
            return h.download(_native.BUF_LINK_POS)
        ptr, nbytes = h.get_buffer(_native.BUF_LINK_POS)           # also brings the rows up to date on the current stream
        if getattr(self, '_lpos_view', None) is None or self._lpos_view[0] != (ptr, nbytes):
            shape = (self.num_envs, self.num_links, 4)
            holder = SimpleNamespace(__cuda_array_interface__={'shape': shape, 'typestr': '<f4', 'data': (ptr, False), 'version': 2,
                                                              'strides': None})
            self._lpos_view = ((ptr, nbytes), torch.as_tensor(holder, device=self.device))
        return self._lpos_view[1]

    def action_buffer(self):
        """The bound int32 [B, num_agents] action tensor (step() with no copy also accepts any contiguous int32 CUDA
        tensor of that shape)."""
        return self._t.get('actions')

    def status_flags(self) -> int:
        return self.simulator.handle.status_flags()

    def close(self) -> None:
        self.simulator.handle.close()
