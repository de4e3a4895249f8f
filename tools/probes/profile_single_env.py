import cProfile, pstats, sys, io
sys.path.insert(0, '.')
import numpy as np
from gym_d2d_amd.envs import D2DEnv
env = D2DEnv({})
obs = env.reset()
rng = np.random.default_rng(0)
acts = [{k: int(rng.integers(0, env.action_space['due' if k.startswith('due') else 'cue'].n)) for k in obs} for _ in range(8)]
for k in range(50): env.step(acts[k % 8])
pr = cProfile.Profile()
pr.enable()
for k in range(2000): env.step(acts[k % 8])
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(18)
print(s.getvalue()[:3500])
