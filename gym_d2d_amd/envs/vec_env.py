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
                 cue_actions: str = 'agent', use_torch: Optional[bool] = None, first_env: int = 0) -> None:
        """cue_actions: 'agent' - step() takes actions for CUEs and DUEs [B, C+P] (reference behaviour);
        'traffic' - CUE links follow the env's traffic model (round-robin RB at max power,
        traffic_model.py:15-22) and step() takes DUE actions only [B, P].
        first_env: global index of env 0 when one logical batch is sharded over several GPUs."""
        env_config = dict(env_config or {})
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
        h.set_env_offset(first_env)
        h.set_obs_mode(self.obs_fn.native_mode)
        h.set_reward(rid, float(getattr(self.reward_fn, 'native_param', 0.0)))
        self.use_torch = (torch is not None and torch.cuda.is_available()) if use_torch is None else use_torch
        self._t = {}
        if self.use_torch:
            self._bind_torch_buffers()
        self.num_steps = 0
        self._episode = 0
        self._seed = cfg.seed if cfg.seed is not None else 0
        if cue_actions == 'traffic':
            rb, pwr = sim.traffic_model.assignments(sim.devices)
            self._cue_raw = (rb.astype(np.int64) * self.num_pwr_actions[self._cue_kind] + pwr).astype(np.int32)

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
        for name, which in (('actions', _native.BUF_ACTIONS), ('rb', _native.BUF_RB), ('pwr', _native.BUF_PWR)):
            self._t[name] = alloc(which, (b * cap,), torch.int32)[:b * n].view(b, n)
        for name, which in _OUTPUTS:
            self._t[name] = alloc(which, (b * cap,), torch.float32)[:b * n].view(b, n)
        self._t['table'] = alloc(_native.BUF_OBS_TABLE, (b * cap * 6,), torch.float32)[:b * n * 6].view(b, n, 6)
        self._t['env_flags'] = alloc(_native.BUF_ENV_FLAGS, (b,), torch.int32)
        if self.obs_fn.native_mode == _native.OBS_LINEAR:
            self._t['obs'] = alloc(_native.BUF_OBS, (b, n, 6 * n), torch.float32)
        self._stream_ptr = None
        self._follow_torch_stream()

    def _follow_torch_stream(self) -> None:
        """Run the library's kernels on torch's CURRENT stream (the null stream by default) so they are ordered with
        the torch ops that produce actions / consume results.  Re-bound only when the caller switches streams."""
        ptr = torch.cuda.current_stream(self.device).cuda_stream
        if ptr != self._stream_ptr:
            self.simulator.handle.set_stream(ptr)
            self._stream_ptr = ptr

    def _view(self) -> SimpleNamespace:
        sim = self.simulator
        if self.use_torch:
            v = dict(self._t)
        else:
            v = {name: sim.fetch(which) for name, which in _OUTPUTS if name != 'reward' or
                 getattr(self.reward_fn, 'native_id', 0)}
            v['rb'] = sim.fetch(_native.BUF_RB); v['pwr'] = sim.fetch(_native.BUF_PWR)
            v['table'] = sim.fetch(_native.BUF_OBS_TABLE)
            v['pos_x'] = sim.fetch(_native.BUF_POS_X); v['pos_y'] = sim.fetch(_native.BUF_POS_Y)
            if self.obs_fn.native_mode == _native.OBS_LINEAR:
                v['obs'] = sim.fetch(_native.BUF_OBS)
        v.update(link_tx=sim.link_tx, link_rx=sim.link_rx, link_type=sim.link_type)
        return SimpleNamespace(**v)

    # ------------------------------------------------------------------ gym-like API
    def reset(self, seed: Optional[int] = None):
        """New positions for every env (device-side sampler), then one step with uniformly random actions on every
        CUE uplink and DUE sidelink to produce the initial SINRs (d2d_env.py:45-60)."""
        if seed is not None:
            self._seed, self._episode = int(seed), 0
        self.num_steps = 0
        if self.use_torch:
            self._follow_torch_stream()
        self.simulator.reset_device(self._seed, self._episode)
        self._episode += 1
        n_cue = self.config.num_rbs * self.num_pwr_actions['cue']
        n_due = self.config.num_rbs * self.num_pwr_actions['due']
        if self.use_torch:
            g = torch.Generator(device=self.device)
            g.manual_seed((self._seed * 1000003 + self._episode) & 0x7FFFFFFFFFFF)
            a = self._t['actions']
            if self.num_cues and self.cue_actions == 'traffic':
                a[:, :self.num_cues] = self._cue_raw_tensor()           # CUE links always follow the traffic model
            elif self.num_cues:
                a[:, :self.num_cues] = torch.randint(0, n_cue, (self.num_envs, self.num_cues), generator=g,
                                                     device=self.device, dtype=torch.int32)
            if self.num_due_pairs:
                a[:, self.num_cues:] = torch.randint(0, n_due, (self.num_envs, self.num_due_pairs), generator=g,
                                                     device=self.device, dtype=torch.int32)
            self.simulator.handle.step()
        else:
            rng = np.random.default_rng((self._seed, self._episode))
            cue = (np.tile(self._cue_raw, (self.num_envs, 1)) if self.cue_actions == 'traffic'
                   else rng.integers(0, n_cue, (self.num_envs, self.num_cues), dtype=np.int32))
            a = np.concatenate([cue, rng.integers(0, n_due, (self.num_envs, self.num_due_pairs), dtype=np.int32)], axis=1)
            self.simulator.step_arrays(a)
        return self._observe(self._view())

    def step(self, actions):
        """actions: int [B, num_agents] (torch CUDA tensor, or NumPy).  Returns (obs, rewards[B,N], dones[B], info)."""
        sim = self.simulator
        if self.use_torch:
            self._follow_torch_stream()
            a = self._t['actions']
            src = actions if torch.is_tensor(actions) else torch.as_tensor(np.asarray(actions), device=self.device)
            if tuple(src.shape) != (self.num_envs, self.num_agents):
                raise ValueError(f'actions must be [{self.num_envs},{self.num_agents}], got {tuple(src.shape)}')
            if self.cue_actions == 'traffic':
                a[:, :self.num_cues] = self._cue_raw_tensor()
                a[:, self.num_cues:] = src
            elif src.data_ptr() != a.data_ptr():
                a.copy_(src)
            sim.handle.step()
        else:
            src = np.asarray(actions, dtype=np.int32)
            if src.shape != (self.num_envs, self.num_agents):
                raise ValueError(f'actions must be [{self.num_envs},{self.num_agents}], got {src.shape}')
            if self.cue_actions == 'traffic':
                src = np.concatenate([np.tile(self._cue_raw, (self.num_envs, 1)), src], axis=1)
            sim.step_arrays(src)
        self.num_steps += 1
        view = self._view()
        obs = self._observe(view)
        rewards = view.reward if getattr(self.reward_fn, 'native_id', 0) else self.reward_fn.compute(view)
        done = self.num_steps >= EPISODE_LENGTH
        dones = (torch.full((self.num_envs,), done, dtype=torch.bool, device=self.device) if self.use_torch
                 else np.full(self.num_envs, done))
        info = {'rb': view.rb, 'tx_pwr_dbm': view.pwr, 'snr_db': view.snr_db, 'sinr_db': view.sinr_db,
                'rate_bps': view.rate_bps, 'capacity_mbps': view.capacity_mbps}
        return obs, rewards, dones, info

    def _cue_raw_tensor(self):
        if not hasattr(self, '_cue_raw_t'):
            self._cue_raw_t = torch.as_tensor(self._cue_raw, device=self.device)
        return self._cue_raw_t

    def _observe(self, view):
        if isinstance(self.obs_fn, ArrayObsFunction):
            return self.obs_fn.compute(view)
        return view.obs

    def action_buffer(self):
        """The bound int32 [B, N] action tensor: writing actions straight into it avoids the copy in step()."""
        return self._t.get('actions')

    def status_flags(self) -> int:
        return self.simulator.handle.status_flags()

    def close(self) -> None:
        self.simulator.handle.close()
