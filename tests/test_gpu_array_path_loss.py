"""GPU tests of the array-native PathLoss plugin (gym_d2d_amd.path_loss.ArrayPathLoss, d2d_set_path_loss_link_table_dev): the
batched counterpart of the reference's PathLoss contract (path_loss.py:12-25; examples/custom_path_loss.py:8-16).  A per-object
__call__ over a batch is B x N x N Python calls per reset; compute(view) evaluates the model once on the GPU and the library
takes the [B,N,N] dB tensor from device memory."""
import math
import time

import numpy as np
import pytest

from golden_util import rel_err
from oracle import d2d_oracle as orc
from sim_util import default_links, random_layout

pytestmark = pytest.mark.gpu
TOL = 1e-5
FIELDS = (('sinr_db', 'BUF_SINR_DB'), ('snr_db', 'BUF_SNR_DB'), ('rate_bps', 'BUF_RATE_BPS'), ('capacity_mbps', 'BUF_CAPACITY'))


def _two_slope_classes():
    from gym_d2d_amd.path_loss import ArrayPathLoss, PathLoss

    class TwoSlopeArray(ArrayPathLoss):            # not a single power law: cannot be lowered to columns
        def compute(self, view):
            xp, d = view.xp, view.distance()
            base = 40.0 + 20.0 * xp.log10(d)
            far = base + 15.0 * xp.log10(d / 50.0) + 0.5 * (view.tx_column(lambda t: t.antenna_height_m) - view.rx_column(lambda r: r.antenna_height_m))
            return xp.where(d < 50.0, base, far)

    class TwoSlopeObject(PathLoss):                # the same model the reference's way: one call per pair
        def __call__(self, tx, rx):
            d = tx.position.distance(rx.position)
            base = 40.0 + 20.0 * math.log10(d)
            return base if d < 50.0 else base + 15.0 * math.log10(d / 50.0) + 0.5 * (tx.antenna_height_m - rx.antenna_height_m)
    return TwoSlopeArray, TwoSlopeObject


def _oracle_table(pos64, cols):
    d = np.hypot(pos64[:, :, None, 0] - pos64[:, None, :, 0], pos64[:, :, None, 1] - pos64[:, None, :, 1])
    with np.errstate(divide='ignore'):
        base = 40.0 + 20.0 * np.log10(d)
        return np.where(d < 50.0, base, base + 15.0 * np.log10(d / 50.0) + 0.5 * (cols.ant_h_m[None, :, None] - cols.ant_h_m[None, None, :]))


def _raw(rng, b, rbs, cues, dues):
    return np.concatenate([rng.integers(0, rbs * 24, (b, cues)), rng.integers(0, rbs * 21, (b, dues))], 1).astype(np.int32)


def test_array_plugin_matches_the_per_object_route_and_the_oracle(native):
    from gym_d2d_amd.simulator import Simulator
    arr_cls, obj_cls = _two_slope_classes()
    b, cues, dues, rbs = 6, 9, 11, 4
    rng = np.random.default_rng(81)
    pos = random_layout(rng, b, cues, dues)
    raw = _raw(rng, b, rbs, cues, dues)
    outs = {}
    for name, cls in (('array', arr_cls), ('object', obj_cls)):
        sim = Simulator(dict(num_rbs=rbs, num_cues=cues, num_due_pairs=dues, num_envs=b, path_loss_model=cls))
        sim.set_positions(pos)
        sim.set_links(sim.default_link_keys())
        sim.step_arrays(raw)
        assert sim.check_flags() & native.FLAG_ZERO_DISTANCE == 0
        outs[name] = {f: sim.fetch(getattr(native, buf)).copy() for f, buf in FIELDS}
        sim.handle.close()
    ids, cfgs, is_bs = orc.device_configs(cues, dues)
    cols = orc.device_columns(cfgs, is_bs)
    tx, rx, ty = default_links(cues, dues)
    p64 = pos.astype(np.float64)
    ref = orc.full_step(p64, tx, rx, ty, raw, cols, orc.PathLossSpec('table', 2.1, table_db=_oracle_table(p64, cols)), with_obs=False)
    for f, _ in FIELDS:
        assert rel_err(outs['array'][f], ref[f]) <= TOL, f
        # the two routes round the same double-precision gain to float32 (pow on the host, exp2 on the device: the last bit of the double may differ)
        assert rel_err(outs['array'][f], outs['object'][f]) <= 1e-6, f


def test_array_plugin_follows_device_resets_link_changes_and_float64_positions(native):
    """Evaluated again after every reset (positions move) and every link-list change (the table is indexed by link); host-supplied
    float64 layouts reach compute() unrounded."""
    from gym_d2d_amd.simulator import Simulator
    arr_cls, _ = _two_slope_classes()
    b, cues, dues, rbs = 4, 6, 6, 3
    sim = Simulator(dict(num_rbs=rbs, num_cues=cues, num_due_pairs=dues, num_envs=b, path_loss_model=arr_cls))
    sim.set_links(sim.default_link_keys())
    ids, cfgs, is_bs = orc.device_configs(cues, dues)
    cols = orc.device_columns(cfgs, is_bs)
    tx, rx, ty = default_links(cues, dues)
    rng = np.random.default_rng(3)
    raw = _raw(rng, b, rbs, cues, dues)
    for episode in range(2):
        sim.reset_device(seed=5, episode=episode)
        p64 = sim.positions().astype(np.float64)
        sim.step_arrays(raw)
        ref = orc.full_step(p64, tx, rx, ty, raw, cols, orc.PathLossSpec('table', 2.1, table_db=_oracle_table(p64, cols)), with_obs=False)
        assert rel_err(sim.fetch(native.BUF_SINR_DB), ref['sinr_db']) <= TOL, episode
    # a DUE-only link list: the table is re-evaluated for it
    keys = list(sim.devices.dues.keys())
    sim.set_links(keys)
    sim.step_arrays(raw[:, cues:])
    ref = orc.full_step(p64, tx[cues:], rx[cues:], ty[cues:], raw[:, cues:], cols, orc.PathLossSpec('table', 2.1, table_db=_oracle_table(p64, cols)),
                        with_obs=False)
    assert rel_err(sim.fetch(native.BUF_SINR_DB), ref['sinr_db']) <= TOL
    sim.set_links([])                                         # no links: nothing to evaluate, nothing raised
    sim.set_links(sim.default_link_keys())
    pos64, _ = orc.sample_positions_from_uniforms(rng.random((b, 1 + cues + 2 * dues, 32, 2)), cues, dues, 500.0, 20.0)
    sim.set_positions(pos64)                                  # float64, not float32-representable
    sim.step_arrays(raw)
    ref = orc.full_step(pos64, tx, rx, ty, raw, cols, orc.PathLossSpec('table', 2.1, table_db=_oracle_table(pos64, cols)), with_obs=False)
    assert rel_err(sim.fetch(native.BUF_SINR_DB), ref['sinr_db']) <= 3e-6
    sim.handle.close()


def test_array_plugin_serves_the_single_env_through_its_derived_call(native):
    """D2DEnv (one env, dict in / dict out) calls model(tx, rx) per pair as the reference does; ArrayPathLoss derives that call from
    compute(), so one definition serves both."""
    from gym_d2d_amd.envs import D2DEnv
    arr_cls, obj_cls = _two_slope_classes()
    import random
    res = {}
    for name, cls in (('array', arr_cls), ('object', obj_cls)):
        random.seed(4)
        env = D2DEnv({'num_rbs': 5, 'num_cues': 6, 'num_due_pairs': 6, 'path_loss_model': cls})
        obs = env.reset()
        _, _, _, info = env.step({k: 7 for k in obs})
        res[name] = np.array([info[k]['sinr_db'] for k in obs])
        env.simulator.handle.close()
    assert rel_err(res['array'], res['object']) <= 1e-6
    m = arr_cls(2.1)
    from gym_d2d_amd.simulator import create_devices
    from gym_d2d_amd.envs.env_config import EnvConfig
    devs = list(create_devices(EnvConfig(num_cues=1, num_due_pairs=1)).values())
    with pytest.raises(ValueError, match='math domain error'):
        m(devs[1], devs[1])                                   # distance 0: what math.log10(0) raises in a per-object model


def test_array_plugin_rejects_a_wrong_shape_and_accepts_float32(native):
    from gym_d2d_amd.path_loss import ArrayPathLoss
    from gym_d2d_amd.simulator import Simulator

    class Wrong(ArrayPathLoss):
        def compute(self, view):
            return view.distance()[:, :, :-1]

    class Single(ArrayPathLoss):
        def compute(self, view):
            return (20.0 * view.xp.log10(view.distance()) + 38.0).float()
    sim = Simulator(dict(num_rbs=3, num_cues=3, num_due_pairs=3, num_envs=2, path_loss_model=Wrong))
    sim.set_links(sim.default_link_keys())
    with pytest.raises(ValueError, match=r'\[2,6,6\]'):
        sim.reset_device(seed=1)
    sim.handle.close()
    b, cues, dues, rbs = 5, 4, 4, 2
    sim = Simulator(dict(num_rbs=rbs, num_cues=cues, num_due_pairs=dues, num_envs=b, path_loss_model=Single))
    sim.set_links(sim.default_link_keys())
    sim.reset_device(seed=2)
    rng = np.random.default_rng(1)
    raw = _raw(rng, b, rbs, cues, dues)
    sim.step_arrays(raw)
    p64 = sim.positions().astype(np.float64)
    d = np.hypot(p64[:, :, None, 0] - p64[:, None, :, 0], p64[:, :, None, 1] - p64[:, None, :, 1])
    with np.errstate(divide='ignore'):
        table = 20.0 * np.log10(d) + 38.0
    ids, cfgs, is_bs = orc.device_configs(cues, dues)
    tx, rx, ty = default_links(cues, dues)
    ref = orc.full_step(p64, tx, rx, ty, raw, orc.device_columns(cfgs, is_bs), orc.PathLossSpec('table', 2.1, table_db=table), with_obs=False)
    assert rel_err(sim.fetch(native.BUF_SINR_DB), ref['sinr_db']) <= 2e-5      # float32 dB near 100 dB carry 3.8e-6 dB of their own
    sim.handle.close()


def test_device_table_entry_validates_its_arguments(native):
    import torch
    h = native.Handle(num_envs=2, num_rbs=4, num_cues=2, num_due_pairs=2, pwr_levels_due=21, pwr_levels_cue=24, pwr_levels_mbs=47)
    t = torch.zeros(2, 2, 2, dtype=torch.float64, device='cuda')
    with pytest.raises(native.NativeError, match='d2d_set_links'):
        h.set_path_loss_link_table_dev(t.data_ptr(), native.F64, 2, True)
    h.set_links([1, 3], [0, 4], [1, 3])
    with pytest.raises(native.NativeError, match='null'):
        h.set_path_loss_link_table_dev(0, native.F64, 2, True)
    with pytest.raises(native.NativeError, match='dtype'):
        h.set_path_loss_link_table_dev(t.data_ptr(), 7, 2, True)
    with pytest.raises(native.NativeError, match='n_links'):
        h.set_path_loss_link_table_dev(t.data_ptr(), native.F64, 3, True)
    h.set_path_loss_link_table_dev(t.data_ptr(), native.F64, 2, True)
    h.close()


def test_array_plugin_installs_a_full_size_batch_in_under_a_second(native):
    """BASELINE config 4 sizes: 4096 envs x 512 links = 1.07e9 table entries per reset.  The per-object route would make 1.1e9
    Python calls and hold an 8.6 GB host table; the array route evaluates on the GPU and hands the tensor over in place."""
    import torch
    from gym_d2d_amd.envs import VecD2DEnv
    from gym_d2d_amd.envs.obs_fn import OwnLinkObsFunction
    arr_cls, _ = _two_slope_classes()
    b, c, p, r = 4096, 256, 256, 256
    env = VecD2DEnv({'num_rbs': r, 'num_cues': c, 'num_due_pairs': p, 'obs_fn': OwnLinkObsFunction, 'path_loss_model': arr_cls}, num_envs=b)
    env.reset(seed=7)                                         # first reset: allocations
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    env.reset()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f'\nArrayPathLoss reset at {b} x {c + p}: {dt * 1e3:.0f} ms (sampler + compute + dB -> gain + the reset step)')
    assert dt < 1.0, dt
    act = torch.randint(0, r * 21, (b, c + p), device=env.device, dtype=torch.int32)
    env.step(act)
    torch.cuda.synchronize()
    sim = env.simulator
    pos = sim.positions()
    ids, cfgs, is_bs = orc.device_configs(c, p)
    cols = orc.device_columns(cfgs, is_bs)
    tx, rx, ty = default_links(c, p)
    pick = [0, 1777, b - 1]
    p64 = pos[pick].astype(np.float64)
    raw = act[pick].cpu().numpy()
    ref = orc.full_step(p64, tx, rx, ty, raw, cols, orc.PathLossSpec('table', 2.1, table_db=_oracle_table(p64, cols)), with_obs=False, chunk=1)
    for f, buf in FIELDS:
        got = sim.fetch(getattr(native, buf))[pick]
        assert rel_err(got, ref[f]) <= TOL, f
    env.close()
