"""D2DEnv - drop-in for gym_d2d.envs.D2DEnv (gym_d2d/envs/d2d_env.py:21-134) running on the MI355X library.

Same constructor (`env_config` dict, popped for 'obs_fn' / 'reward_fn' classes), same reset()/step()/render()/
save_device_config(), same dict-in / dict-out multi-agent convention with 'tx_id:rx_id' keys and the old gym 4-tuple.
One env == a batch of 1 on the GPU; use VecD2DEnv for thousands of envs.
"""
from __future__ import annotations

import json
from pathlib import Path
from typing import Any, Dict, Tuple

import numpy as np

from .. import _native
from ..actions import Action, Actions, make_action
from ..id import Id
from ..simulator import BASE_STATION_ID, NativeState, Simulator
from ..spaces import Dict as DictSpace
from ..spaces import Discrete, Env
from .obs_fn import LinearObsFunction, ObsFunction
from .reward_fn import RewardFunction, SystemCapacityRewardFunction

EPISODE_LENGTH = 10
DEFAULT_OBS_FN = LinearObsFunction
DEFAULT_REWARD_FN = SystemCapacityRewardFunction


class D2DEnv(Env):
    metadata = {'render.modes': ['human']}

    def __init__(self, env_config=None) -> None:
        super().__init__()
        env_config = env_config or {}
        # NB like the reference this pops from (mutates) the caller's dict
        self.obs_fn: ObsFunction = env_config.pop('obs_fn', DEFAULT_OBS_FN)()
        self.reward_fn: RewardFunction = env_config.pop('reward_fn', DEFAULT_REWARD_FN)()
        if int(env_config.get('num_envs', 1)) != 1:
            raise ValueError('D2DEnv is the single-env API; use VecD2DEnv for num_envs > 1')
        self.simulator = Simulator(env_config)
        cfg = self.simulator.config
        self.observation_space = self.obs_fn.get_obs_space(cfg)
        self.num_pwr_actions = cfg.num_pwr_actions
        self.action_space = DictSpace({
            kind: Discrete(cfg.num_rbs * levels) for kind, levels in self.num_pwr_actions.items()
        })
        self.actions = None
        self.state = None
        self.num_steps = 0
        self._key_cache: Dict[str, tuple] = {}     # 'tx:rx' -> (ids, tx device, rx device, link type, power levels)
        # plugin lowering: built-ins run inside the kernels, anything else is Python over the GPU results
        h = self.simulator.handle
        self._native_obs = isinstance(self.obs_fn, LinearObsFunction) and \
            type(self.obs_fn).get_state is LinearObsFunction.get_state
        h.set_obs_mode(_native.OBS_LINEAR if self._native_obs else _native.OBS_TABLE)
        rid = getattr(self.reward_fn, 'native_id', _native.REWARD_NONE)
        self._native_reward = rid != _native.REWARD_NONE and self._reward_call_is_builtin()
        if self._native_reward:
            h.set_reward(rid, float(self.reward_fn.native_param))
            self._reward_key = (rid, float(self.reward_fn.native_param))
        else:
            h.set_reward(_native.REWARD_NONE, 0.0)
            self._reward_key = None

    def _reward_call_is_builtin(self) -> bool:
        for klass in type(self.reward_fn).__mro__:
            if '__call__' in klass.__dict__:
                return klass.__module__ == 'gym_d2d_amd.envs.reward_fn'
        return False

    # ------------------------------------------------------------------ gym API
    def reset(self):
        self.num_steps = 0
        self.simulator.reset()
        # one step with random actions on every CUE uplink and DUE sidelink gives the initial SINRs
        self.actions = self._reset_random_actions()
        self.state = self._run(self.actions)
        return self.obs_fn.get_state(self.actions, self.state, self.simulator.devices)

    def step(self, raw_actions: Dict[str, Any]):
        self.actions = self._extract_actions(raw_actions)
        self.state = self._run(self.actions)
        self.num_steps += 1
        obs = self.obs_fn.get_state(self.actions, self.state, self.simulator.devices)
        rewards = self.reward_fn(self.actions, self.state)
        game_over = {'__all__': self.num_steps >= EPISODE_LENGTH}
        info = self._infos(self.actions, self.state)
        return obs, rewards, game_over, info

    def render(self, mode='human'):
        assert self.state is not None and self.actions is not None, \
            'Initialise environment with `reset()` before calling `render()`'
        print(self.obs_fn.get_state(self.actions, self.state, self.simulator.devices))

    def save_device_config(self, config_file: Path) -> None:
        """Write {device id: {'position': (x, y), 'config': {...}}} as JSON (same format as d2d_env.py:124-134);
        load it back with env_config['device_config_file']."""
        snapshot = {}
        for device in self.simulator.devices.values():
            snapshot[device.id] = {'position': device.position.as_tuple(), 'config': device.config}
        with config_file.open(mode='w') as fid:
            json.dump(snapshot, fid)

    def close(self) -> None:
        """Release the GPU-side state (gym.Env.close)."""
        self.simulator.handle.close()

    # ------------------------------------------------------------------ internals
    def _run(self, actions: Actions) -> NativeState:
        state = self.simulator.step(actions)          # every result of the step arrives in one packed host block
        if not self._native_obs:
            state.linear_obs = None
        if self._native_reward:
            state.native_reward_key = self._reward_key
        else:
            state.native_reward = None
        return state

    def _reset_random_actions(self) -> Actions:
        devs = self.simulator.devices
        acts = Actions()
        for tx_id in devs.cues.keys():
            acts[(tx_id, BASE_STATION_ID)] = self._extract_action(tx_id, BASE_STATION_ID, self.action_space['cue'].sample())
        for tx_id, rx_id in devs.dues.keys():
            acts[(tx_id, rx_id)] = self._extract_action(tx_id, rx_id, self.action_space['due'].sample())
        return acts

    def _extract_actions(self, raw_actions: Dict[str, Any]) -> Actions:
        """{'tx:rx': raw} -> Actions (d2d_env.py:73-101).  Key parsing, link typing and the device lookups are cached
        per key string - agents' keys repeat every step."""
        acts = Actions()
        data, cache = acts.data, self._key_cache
        for pair, raw in raw_actions.items():
            ent = cache.get(pair)
            if ent is None:
                ids = tuple(Id(part) for part in pair.split(':'))
                if len(ids) != 2:                                # the reference's _extract_action(*ids, raw) arity error
                    raise TypeError(f"action key must be 'tx_id:rx_id', got {pair!r}")
                link_type, kind = self.simulator.classify(ids[0])
                devs = self.simulator.devices
                ent = cache[pair] = (ids, devs[ids[0]], devs[ids[1]], link_type, self.num_pwr_actions[kind], kind)
            ids, tx, rx, link_type, levels, kind = ent
            if type(raw) is int:
                rb, pwr = divmod(raw, levels)
            else:
                rb, pwr = self._decode_action(raw, kind)
            data[ids] = make_action(tx, rx, link_type, rb, pwr)
        return acts

    def _extract_action(self, tx_id: Id, rx_id: Id, action: Any) -> Action:
        link_type, kind = self.simulator.classify(tx_id)
        rb, pwr = self._decode_action(action, kind)
        devs = self.simulator.devices
        return Action(devs[tx_id], devs[rx_id], link_type, rb, pwr)

    def _decode_action(self, action: Any, tx_type: str) -> Tuple[int, int]:
        """int -> (a // P, a % P); ndarray of ndim 2 -> (rb, pwr) given directly (d2d_env.py:93-101)."""
        if isinstance(action, (int, np.integer)):
            rb, pwr = divmod(int(action), self.num_pwr_actions[tx_type])
        elif isinstance(action, np.ndarray) and action.ndim == 2:
            rb, pwr = (np.asarray(part) for part in action)
            if rb.size != 1 or pwr.size != 1:
                raise TypeError('only size-1 arrays can be converted to Python scalars')
            rb, pwr = rb.item(), pwr.item()
        else:
            raise ValueError(f'Unable to decode action type "{type(action)}"')
        return int(rb), int(pwr)

    def _infos(self, actions: Actions, state: dict) -> Dict[str, Any]:
        lists = getattr(state, 'lists', None)
        if lists is None:
            return {':'.join(ids): self._info(act, state) for ids, act in actions.items()}
        # same six keys as _info, read positionally from the step's result arrays (agent order == actions' order)
        snr, sinr, rate, cap = lists['snrs_db'], lists['sinrs_db'], lists['rate_bps'], lists['capacity_mbps']
        return {':'.join(ids): {'rb': act.rb, 'tx_pwr_dbm': act.tx_pwr_dBm, 'snr_db': snr[k], 'sinr_db': sinr[k],
                                'rate_bps': rate[k], 'capacity_mbps': cap[k]}
                for k, (ids, act) in enumerate(getattr(actions, 'data', actions).items())}

    def _info(self, action: Action, state: dict) -> Dict[str, Any]:
        ids = (action.tx.id, action.rx.id)
        return {
            'rb': action.rb,
            'tx_pwr_dbm': action.tx_pwr_dBm,
            'snr_db': state['snrs_db'][ids],
            'sinr_db': state['sinrs_db'][ids],
            'rate_bps': state['rate_bps'][ids],
            'capacity_mbps': state['capacity_mbps'][ids],
        }
