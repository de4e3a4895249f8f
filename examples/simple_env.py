"""The reference's agent loop (gym-d2d examples/simple_env.py) on the MI355X implementation.

Every step returns {agent_id: observation}, takes {agent_id: action} and returns rewards / info keyed the same way.
`gym` is optional: gym_d2d_amd.make() builds the same env without it.
"""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))      # run from a checkout without installing

import gym_d2d_amd

env = gym_d2d_amd.make('D2DEnv-v0')

obses = env.reset()
for _ in range(10):
    actions = {}
    for agent_id in obses:
        kind = 'due' if agent_id.startswith('due') else ('cue' if agent_id.startswith('cue') else 'mbs')
        actions[agent_id] = env.action_space[kind].sample()      # or agent.act(obses[agent_id])
    obses, rewards, game_over, info = env.step(actions)
env.render()
print('episode finished:', game_over, '| reward', next(iter(rewards.values())))
