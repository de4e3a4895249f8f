"""Fix device positions / per-device link budgets across experiments through the reference's JSON format
(gym-d2d examples/saving_loading_device_config.py)."""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))      # run from a checkout without installing

import tempfile
from pathlib import Path

import gym_d2d_amd

env = gym_d2d_amd.make('D2DEnv-v0')
env.reset()                                            # draws random positions
path = Path(tempfile.mkdtemp()) / 'device_config.json'
env.save_device_config(path)

env2 = gym_d2d_amd.make('D2DEnv-v0', env_config={'device_config_file': path})
env2.reset()                                           # every device listed in the file keeps its position
same = all(env.simulator.devices[i].position == env2.simulator.devices[i].position for i in env.simulator.devices)
print('positions restored from', path.name, ':', same)
