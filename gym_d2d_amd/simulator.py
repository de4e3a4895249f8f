"""Host side of the simulator (mirrors gym_d2d/simulator.py) over the HIP library.

`Simulator` keeps the reference's surface - .config, .devices, .traffic_model, .path_loss, reset(), step(actions) ->
{'sinrs_db', 'snrs_db', 'rate_bps', 'capacity_mbps'} - but owns a d2d_handle with B = config.num_envs environments
whose state (positions, actions, outputs) lives in HBM.  All arithmetic of Simulator.step (simulator.py:77-154)
happens in csrc/d2d_step.hip; this file only lowers configuration to tables and moves arrays.  There is no host
implementation of the step: without libd2d_hip.so and a gfx950 GPU, construction raises.
"""
from __future__ import annotations

import random
from typing import Dict, Iterable, List, Optional, Sequence, Tuple

import numpy as np

from . import _native
from .actions import Actions
from .device import BaseStation, UserEquipment, link_budget_columns
from .devices import Devices
from .envs.env_config import EnvConfig
from .id import Id
from .link_type import LinkType
from .path_loss import ArrayPathLoss, PathLoss, PathLossView
from .position import Position, get_random_position, get_random_position_nearby
from .traffic_model import TrafficModel

BASE_STATION_ID = Id('mbs')


def create_devices(config: EnvConfig) -> Devices:
    """One base station 'mbs', CUEs 'cue00'.., DUE pairs ('due00','due01'), ('due02','due03').. (simulator.py:18-50).
    A device listed in the device_config_file takes its 'config' from there instead of the env-level defaults."""
    shared = {'num_subcarriers': config.num_subcarriers, 'subcarrier_spacing_kHz': config.subcarrier_spacing_kHz}
    by_class = {
        'mbs': shared,
        'cue': {**shared, 'max_tx_power_dBm': config.cue_max_tx_power_dBm},
        'due': {**shared, 'max_tx_power_dBm': config.due_max_tx_power_dBm},
    }

    def cfg_for(dev_id: str) -> dict:
        return config.devices.get(dev_id, {}).get('config', by_class[dev_id[:3]])

    bs = BaseStation(BASE_STATION_ID, cfg_for(BASE_STATION_ID))
    cues = {}
    for k in range(config.num_cues):
        cue_id = Id(f'cue{k:02d}')
        cues[cue_id] = UserEquipment(cue_id, cfg_for(cue_id))
    dues = {}
    for k in range(config.num_due_pairs):
        tx_id, rx_id = Id(f'due{2 * k:02d}'), Id(f'due{2 * k + 1:02d}')
        dues[(tx_id, rx_id)] = (UserEquipment(tx_id, cfg_for(tx_id)), UserEquipment(rx_id, cfg_for(rx_id)))
    return Devices(bs, cues, dues)


class NativeState(dict):
    """The `state` dict of one env, backed by arrays the GPU produced.  The reference's four keys map
    (tx_id, rx_id) -> float; extra attributes carry the kernel outputs built-in plugins re-key."""

    FIELDS = (('sinrs_db', _native.BUF_SINR_DB), ('snrs_db', _native.BUF_SNR_DB), ('rate_bps', _native.BUF_RATE_BPS),
              ('capacity_mbps', _native.BUF_CAPACITY))

    def __init__(self, keys: Sequence[Tuple[Id, Id]], arrays: Dict[str, np.ndarray]) -> None:
        super().__init__()
        self.arrays = arrays
        self.lists = {name: arrays[name].tolist() for name, _ in self.FIELDS}   # Python floats, as the reference's dicts hold
        for name, _ in self.FIELDS:
            self[name] = dict(zip(keys, self.lists[name]))
        self.linear_obs: Optional[np.ndarray] = None
        self.obs_table: Optional[np.ndarray] = None
        self.native_reward: Optional[np.ndarray] = None
        self.native_reward_key = None


class Simulator:
    """B = config.num_envs environments on one GPU.  For B == 1 the object model (self.devices with positions)
    is kept in sync so Python plugins see what the reference would show them."""

    def __init__(self, env_config: dict, *, max_links: Optional[int] = None) -> None:
        self.config = EnvConfig(**env_config)
        self.devices: Devices = create_devices(self.config)
        self.traffic_model: TrafficModel = self.config.traffic_model(self.config.num_rbs)
        self.path_loss: PathLoss = self.config.path_loss_model(self.config.carrier_freq_GHz)
        self.num_envs = int(self.config.num_envs)
        n_default = self.config.num_cues + self.config.num_due_pairs
        # room for every uplink, every downlink and every sidelink at once
        cap = max_links if max_links is not None else min(_native.MAX_LINKS, 2 * self.config.num_cues + self.config.num_due_pairs)
        cap = max(cap, n_default, 1)
        p = self.config.num_pwr_actions
        self.handle = _native.Handle(
            num_envs=self.num_envs, num_rbs=self.config.num_rbs, num_cues=self.config.num_cues,
            num_due_pairs=self.config.num_due_pairs, pwr_levels_due=p['due'], pwr_levels_cue=p['cue'],
            pwr_levels_mbs=p['mbs'], max_links=cap, device_ordinal=self.config.device_ordinal,
            cell_radius_m=self.config.cell_radius_m, d2d_radius_m=self.config.d2d_radius_m)
        self._dev_list = list(self.devices.values())
        self._link_keys: List[Tuple[Id, Id]] = []
        self._link_sig = None
        self._episode = 0
        self._table_route = False
        self._pl_covered = 0                         # 0: no positions yet; None: positions set, table not evaluated
        self._pl_positions = None
        self._install_tables()

    # ------------------------------------------------------------------ lowering of configuration
    def _install_tables(self) -> None:
        cols = link_budget_columns(self._dev_list)
        self.handle.set_device_table(cols['eirp_off_db'], cols['rx_off_db'], cols['noise_dbm'], cols['sens_dbm'],
                                     cols['bw_hz'])
        law = self.path_loss.power_law_columns(self._dev_list)
        if law is not None:
            shadow = law.get('shadowing')
            if shadow:
                seed = self.config.seed if self.config.seed is not None else random.getrandbits(63)
                self.handle.set_path_loss_shadowing(law['a_tx_db'], law['a_rx_db'], law['exponent'], shadow['d0_m'],
                                                    shadow['chi_dB'], seed)
                self.shadowing_seed = seed
            else:
                self.handle.set_path_loss_power_law(law['a_tx_db'], law['a_rx_db'], law['exponent'])
            self._table_route = False
        else:
            self._table_route = True    # evaluated per episode in _refresh_path_loss_table()

    def _refresh_path_loss_table(self, positions: Optional[np.ndarray] = None) -> None:
        """Python-plugin route: evaluate the user's PathLoss for the (transmitter of a link) x (receiver of a link) device
        pairs of every env - the only entries the step reads; with no link list yet the evaluation waits for set_links."""
        if not self._table_route:
            return
        self._pl_covered = None                     # positions changed: nothing evaluated for them yet
        # a COPY: the caller may reuse its array before a later set_links re-evaluates the table from it
        self._pl_positions = None if positions is None else np.array(positions, dtype=np.float64, copy=True)
        if self._link_sig:
            self._evaluate_path_loss_table()

    def _evaluate_array_path_loss(self) -> bool:
        """ArrayPathLoss on a batch: compute(view) on the GPU, the [B,N,N] dB tensor handed over from device memory
        (d2d_set_path_loss_link_table_dev).  False when torch / CUDA is not there (the host routes below then serve it)."""
        try:
            import torch
        except Exception:       # pragma: no cover
            return False
        if not torch.cuda.is_available():
            return False
        h = self.handle
        dev = torch.device('cuda', self.config.device_ordinal)
        n = len(self.link_tx)
        with torch.cuda.device(dev):
            if self._pl_positions is not None:               # host-supplied positions, possibly float64: as given
                p = torch.as_tensor(np.asarray(self._pl_positions), device=dev)
                tx, rx = p[:, torch.as_tensor(self.link_tx.astype(np.int64), device=dev)], p[:, torch.as_tensor(self.link_rx.astype(np.int64), device=dev)]
                cols = (tx[..., 0], tx[..., 1], rx[..., 0], rx[..., 1])
            else:                                            # device-side reset: POS_X / POS_Y where they are, gathered per link by torch
                def plane(which):
                    ptr, _ = h.get_buffer(which)
                    holder = type('Plane', (), {'__cuda_array_interface__': {'shape': (self.num_envs, len(self._dev_list)), 'typestr': '<f4',
                                                                             'data': (ptr, False), 'version': 2, 'strides': None}})()
                    return torch.as_tensor(holder, device=dev)
                px, py = plane(_native.BUF_POS_X), plane(_native.BUF_POS_Y)
                h.synchronize()                              # the sampler wrote them on the handle's stream; torch reads on its own
                jt, jr = torch.as_tensor(self.link_tx.astype(np.int64), device=dev), torch.as_tensor(self.link_rx.astype(np.int64), device=dev)
                cols = (px[:, jt], py[:, jt], px[:, jr], py[:, jr])
            view = PathLossView(torch, *cols, [self._dev_list[i] for i in self.link_tx], [self._dev_list[i] for i in self.link_rx], like=cols[0])
            pl = self.path_loss.compute(view)
            if tuple(pl.shape) != (self.num_envs, n, n):
                raise ValueError(f'ArrayPathLoss.compute must return [{self.num_envs},{n},{n}], got {tuple(pl.shape)}')
            if pl.dtype not in (torch.float32, torch.float64):
                pl = pl.double()
            pl = pl.contiguous()
            torch.cuda.current_stream(dev).synchronize()     # the conversion kernel runs on the handle's stream
            h.set_path_loss_link_table_dev(pl.data_ptr(), _native.F64 if pl.dtype == torch.float64 else _native.F32, n, True)
        return True

    def _evaluate_path_loss_table(self) -> None:
        if len(self.link_tx) == 0:
            return                                       # no links, no pairs: the next non-empty list evaluates them
        if self.num_envs > 1 and isinstance(self.path_loss, ArrayPathLoss) and self._evaluate_array_path_loss():
            self._pl_covered = (set(self.link_tx.tolist()), set(self.link_rx.tolist()))
            return
        txs, rxs = set(self.link_tx.tolist()), set(self.link_rx.tolist())
        if self.num_envs == 1:
            # one env: a [D,D] DEVICE table, which survives the link list changing from step to step (D2DEnv steps whatever
            # subset of links the action dict names); same positions + a changed list: only the pairs not yet there are evaluated
            if self._pl_covered is not None:
                if txs <= self._pl_covered[0] and rxs <= self._pl_covered[1]:
                    return
                txs |= self._pl_covered[0]; rxs |= self._pl_covered[1]
            self._pl_table = self.path_loss.table_db(self._dev_list, txs, rxs)
            self.handle.set_path_loss_table(self._pl_table)
            self._pl_covered = (txs, rxs)
            return
        # a batch: [B,N,N] by (tx LINK, rx LINK) - exactly the pairs the step reads (d2d_set_path_loss_link_table), not the
        # dense [B,D,D] device cube (9.7 GB at 4096 x 769 devices).  Tied to the link list: re-evaluated when it changes.
        positions = np.array(self._pl_positions if self._pl_positions is not None else self.positions(), copy=True)
        n = len(self.link_tx)
        tables = np.empty((self.num_envs, n, n), dtype=np.float64)
        pick = np.ix_(self.link_tx, self.link_rx)
        saved = [d.position for d in self._dev_list]
        try:
            for b in range(self.num_envs):
                for d, xy in zip(self._dev_list, positions[b]):
                    d.set_position(Position(float(xy[0]), float(xy[1])))
                tables[b] = self.path_loss.table_db(self._dev_list, txs, rxs)[pick]
        finally:
            for d, p in zip(self._dev_list, saved):
                d.set_position(p)
        self._pl_table = tables
        self.handle.set_path_loss_link_table(tables)
        self._pl_covered = (txs, rxs)

    def fixed_positions(self):
        """(mask[D] uint8, xy[D,2] float64) of devices pinned by the device_config_file (simulator.py:65-66), in the file's own
        precision (JSON numbers are Python floats)."""
        d = len(self._dev_list)
        mask = np.zeros(d, dtype=np.uint8)
        xy = np.zeros((d, 2), dtype=np.float64)
        for k, dev in enumerate(self._dev_list):
            if dev.id != BASE_STATION_ID and dev.id in self.config.devices:
                mask[k] = 1
                xy[k] = self.config.devices[dev.id]['position']
        return mask, xy

    # ------------------------------------------------------------------ links
    def default_link_keys(self) -> List[Tuple[Id, Id]]:
        """All CUE uplinks, then all DUE sidelinks - the agent order reset() produces (d2d_env.py:54-60)."""
        keys = [(cue_id, BASE_STATION_ID) for cue_id in self.devices.cues.keys()]
        keys.extend(self.devices.dues.keys())
        return keys

    def classify(self, tx_id: Id) -> Tuple[LinkType, str]:
        """Link type and transmitter class from the transmitter's id (d2d_env.py:80-91)."""
        if tx_id in self.devices.due_pairs:
            return LinkType.SIDELINK, 'due'
        if tx_id in self.devices.cues:
            return LinkType.UPLINK, 'cue'
        return LinkType.DOWNLINK, 'mbs'

    def set_links(self, keys: Iterable[Tuple[Id, Id]]) -> None:
        """Select which (tx_id, rx_id) pairs act, in agent order.  Cached: re-uploading only on change."""
        sig = tuple(keys)
        if sig == self._link_sig:
            return
        keys = [(Id(t), Id(r)) for t, r in sig]
        sig = tuple(keys)
        tx = [self.devices.index_of(t) for t, _ in keys]        # KeyError for unknown ids, as devices.py:28
        rx = [self.devices.index_of(r) for _, r in keys]
        types = [self.classify(t)[0].value for t, _ in keys]
        self.handle.set_links(tx, rx, types)
        self._link_keys = keys
        self._link_sig = sig
        self.link_tx = np.asarray(tx, dtype=np.int32)
        self.link_rx = np.asarray(rx, dtype=np.int32)
        self.link_type = np.asarray(types, dtype=np.int32)
        if self._table_route and getattr(self, '_pl_covered', 0) != 0:
            self._evaluate_path_loss_table()         # pairs the new link list reads that the table does not hold yet

    @property
    def link_keys(self) -> List[Tuple[Id, Id]]:
        return self._link_keys

    # ------------------------------------------------------------------ reset
    def reset(self) -> None:
        """Draw new device positions (simulator.py:61-75).

        B == 1: sampled on the host with Python's `random`, in the reference's device order, so `random.seed(k)`
        gives the reference's layout - in the reference's precision: the Device objects hold the float64 coordinates and the
        GPU gets them as (hi, lo) float32 pairs (d2d_set_positions_f64).  B > 1: sampled on the GPU in float32
        (csrc/d2d_reset.hip)."""
        if self.num_envs == 1:
            self._reset_host()
        else:
            self.reset_device(self.config.seed if self.config.seed is not None else 0)

    def _reset_host(self) -> None:
        for device in self.devices.values():
            if device.id == BASE_STATION_ID:
                pos = Position(0, 0)
            elif device.id in self.config.devices:
                pos = Position(*self.config.devices[device.id]['position'])
            elif device.id in self.devices.cues or device.id in self.devices.due_pairs:
                pos = get_random_position(self.config.cell_radius_m)
            elif device.id in self.devices.due_pairs_inv:
                anchor = self.devices[self.devices.due_pairs_inv[device.id]]
                pos = get_random_position_nearby(self.config.cell_radius_m, anchor.position, self.config.d2d_radius_m)
            else:
                raise ValueError(f'Invalid configuration for device "{device.id}".')
            device.set_position(pos)
        self.push_positions()

    def push_positions(self) -> None:
        """Upload the Device objects' positions (B == 1) and refresh a host-evaluated path-loss table."""
        xy = np.array([d.position.as_tuple() for d in self._dev_list], dtype=np.float64)
        self.handle.set_positions(np.tile(xy[None, :, 0], (self.num_envs, 1)), np.tile(xy[None, :, 1], (self.num_envs, 1)))
        self._refresh_path_loss_table()

    def set_positions(self, positions: np.ndarray) -> None:
        """positions [B, D, 2] -> HBM; B == 1 also updates the Device objects.  A float64 array is taken in the reference's own
        precision (position.py:7-12; d2d_set_positions_f64: hi + lo float32 pairs on the device, differences exact to ~1e-7 -
        when every value is float32-representable that is the float32 upload, same kernels, same bits); any other dtype is
        uploaded as float32."""
        positions = np.asarray(positions)
        if positions.dtype != np.float64:
            positions = positions.astype(np.float32)
        self.handle.set_positions(positions[..., 0], positions[..., 1])
        if self.num_envs == 1:
            for d, xy in zip(self._dev_list, positions[0]):
                d.set_position(Position(float(xy[0]), float(xy[1])))
        self._refresh_path_loss_table(positions)

    def reset_device(self, seed: int, episode: Optional[int] = None) -> None:
        if episode is None:
            episode = self._episode
            self._episode += 1
        mask, xy = self.fixed_positions()
        if mask.any():
            self.handle.reset_positions(seed, episode, mask, xy)
            if (xy != xy.astype(np.float32)).any():
                # pinned coordinates that float32 cannot hold (a device_config_file the reference saved): the sampled layout comes
                # back once, the pinned devices take the file's float64 values, and the whole goes up as (hi, lo) pairs
                pos = self.positions().astype(np.float64)
                pos[:, mask.astype(bool)] = xy[mask.astype(bool)]
                self.handle.set_positions(pos[..., 0], pos[..., 1])
        else:
            self.handle.reset_positions(seed, episode)
        if self.num_envs == 1:
            pos = self.positions()[0].astype(np.float64)
            pos[mask.astype(bool)] = xy[mask.astype(bool)]
            for d, p in zip(self._dev_list, pos):
                d.set_position(Position(float(p[0]), float(p[1])))
        self._refresh_path_loss_table()

    def positions(self) -> np.ndarray:
        """[B, D, 2] float32 copy of the device positions in HBM."""
        return np.stack([self.handle.download(_native.BUF_POS_X), self.handle.download(_native.BUF_POS_Y)], axis=-1)

    # ------------------------------------------------------------------ step
    def step(self, actions: Actions) -> NativeState:
        """Single-env, object-level entry (simulator.py:77-87): actions -> state dict, computed on the GPU.  One packed
        host -> device copy of (rb, pwr), the kernels, one packed device -> host copy of every result (d2d_step_host)."""
        if self.num_envs != 1:
            raise ValueError('Simulator.step(Actions) is the single-env entry; use step_arrays for batches')
        if len(actions) == 0:
            raise ZeroDivisionError('division by zero')      # what reward_fn.py:42 does with no actions
        self.set_links(actions.keys())
        acts = list(getattr(actions, 'data', actions).values())     # UserDict: the dict's own view, not the abc mixin
        rb = np.fromiter((a.rb for a in acts), dtype=np.int32, count=len(actions))[None]
        pwr = np.fromiter((a.tx_pwr_dBm for a in acts), dtype=np.int32, count=len(actions))[None]
        res = self.handle.step_host(rb, pwr)
        if int(res['env_flags'][0]) & _native.FLAG_ZERO_DISTANCE:
            raise ValueError('math domain error')            # log10(0) in path_loss.py:66
        arrays = {'sinrs_db': res['sinr_db'][0].astype(np.float64), 'snrs_db': res['snr_db'][0].astype(np.float64),
                  'rate_bps': res['rate_bps'][0].astype(np.float64), 'capacity_mbps': res['capacity'][0].astype(np.float64)}
        if self._table_route and not all(np.isfinite(v).all() for v in arrays.values()):
            raise ValueError('math domain error')            # the user's PathLoss could not evaluate a used pair
        state = NativeState(self._link_keys, arrays)
        state.obs_table = res['obs_table'][0].astype(np.float64)
        if 'obs' in res:
            state.linear_obs = res['obs'][0].astype(np.float64)
        state.native_reward = res['reward'][0].astype(np.float64)
        return state

    def step_arrays(self, actions: Optional[np.ndarray] = None, *, rb: Optional[np.ndarray] = None,
                    pwr: Optional[np.ndarray] = None, actions_ptr: int = 0) -> None:
        """Enqueue one step for all B envs.  Either raw int actions [B,N] (host array or device pointer) or
        explicit rb / pwr [B,N] host arrays."""
        h = self.handle
        if actions_ptr:
            h.step(actions_ptr)
        elif actions is not None:
            a = np.ascontiguousarray(actions, dtype=np.int32)
            n_agents = h.num_links - h.num_fixed
            if a.shape != (self.num_envs, n_agents):
                raise ValueError(f'actions must be [{self.num_envs},{n_agents}], got {a.shape}')
            if n_agents:
                h.upload(_native.BUF_ACTIONS, a)
            h.step()
        else:
            r = np.ascontiguousarray(rb, dtype=np.int32); p = np.ascontiguousarray(pwr, dtype=np.int32)
            if r.shape != (self.num_envs, h.num_links) or p.shape != r.shape:
                raise ValueError(f'rb/pwr must be [{self.num_envs},{h.num_links}]')
            h.upload(_native.BUF_RB, r)
            h.upload(_native.BUF_PWR, p)
            h.step_rb_pwr()

    def check_flags(self) -> int:
        """Raise what the reference would have raised for this step; returns the flag word otherwise."""
        flags = self.handle.status_flags()
        if flags & _native.FLAG_ZERO_DISTANCE:
            raise ValueError('math domain error')            # log10(0) in path_loss.py:66
        return flags

    def fetch(self, which: int, env_begin: int = 0, env_count: Optional[int] = None) -> np.ndarray:
        return self.handle.download(which, env_begin, env_count)

    def state_of_env(self, b: int) -> NativeState:
        self.check_flags()
        arrays = {name: self.fetch(buf, b, 1)[0].astype(np.float64) for name, buf in NativeState.FIELDS}
        if self._table_route and not all(np.isfinite(v).all() for v in arrays.values()):
            raise ValueError('math domain error')            # the user's PathLoss could not evaluate a used pair
        return NativeState(self._link_keys, arrays)
