"""A user-defined PathLoss plugin (gym-d2d examples/custom_path_loss.py).  Pass the CLASS in env_config; the env
evaluates it on the host once per episode for every device pair and the GPU kernels consume the resulting table."""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))      # run from a checkout without installing

from math import log10

import gym_d2d_amd
from gym_d2d_amd.device import Device
from gym_d2d_amd.path_loss import PathLoss


class FooPathLoss(PathLoss):
    def __call__(self, tx: Device, rx: Device) -> float:
        d = tx.position.distance(rx.position)
        return 20 * log10(d) - tx.tx_antenna_gain_dBi - rx.rx_antenna_gain_dBi


env = gym_d2d_amd.make('D2DEnv-v0', env_config={'path_loss_model': FooPathLoss})
obses = env.reset()
for _ in range(10):
    actions = {agent_id: env.action_space['due' if agent_id.startswith('due') else 'cue'].sample() for agent_id in obses}
    obses, rewards, game_over, info = env.step(actions)
print('sinr of', next(iter(info)), '=', next(iter(info.values()))['sinr_db'], 'dB')
