"""The array-native form of examples/custom_path_loss.py: the same plugin written once as ArrayPathLoss.compute(view) serves the
single-env D2DEnv (which calls model(tx, rx) per pair, as the reference does - the call is derived from compute) AND a batch, where
compute runs once per reset on the GPU and the library takes the [B, N, N] dB tensor from device memory
(d2d_set_path_loss_link_table_dev): no host table, no B x N x N Python calls."""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))      # run from a checkout without installing

import gym_d2d_amd
from gym_d2d_amd.envs import VecD2DEnv
from gym_d2d_amd.path_loss import ArrayPathLoss


class FooArrayPathLoss(ArrayPathLoss):
    def compute(self, view):
        # view.xp is torch (batch, on the GPU) or numpy (one pair); pl_db[b, j, i] = PL(tx of link j -> rx of link i)
        return (20 * view.xp.log10(view.distance())
                - view.tx_column(lambda tx: tx.tx_antenna_gain_dBi) - view.rx_column(lambda rx: rx.rx_antenna_gain_dBi))


env = gym_d2d_amd.make('D2DEnv-v0', env_config={'path_loss_model': FooArrayPathLoss})
obses = env.reset()
obses, rewards, game_over, info = env.step({a: env.action_space['due' if a.startswith('due') else 'cue'].sample() for a in obses})
print('single env: sinr of', next(iter(info)), '=', next(iter(info.values()))['sinr_db'], 'dB')

venv = VecD2DEnv({'num_rbs': 64, 'num_cues': 64, 'num_due_pairs': 64, 'path_loss_model': FooArrayPathLoss}, num_envs=1024)
venv.reset(seed=1)
obs, rewards, dones, info = venv.step(venv.action_buffer().random_(0, 64 * 21))
print('batch of 1024 x 128 links: mean sinr', float(info['sinr_db'].mean()), 'dB')
