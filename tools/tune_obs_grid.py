import sys, statistics
sys.path.insert(0, ".")
from gym_d2d_amd import _native
from gym_d2d_amd.envs import VecD2DEnv
env = VecD2DEnv({"num_rbs": 256, "num_cues": 256, "num_due_pairs": 256}, num_envs=4096)
env.reset(seed=1); h = env.simulator.handle; act = env.action_buffer()
ref = env._t['obs'][:64].clone()
def timed(n=4):
    h.profile_reset(); h.profile_enable(True)
    for _ in range(n): h.step(act.data_ptr())
    ms, k = h.profile_read(1); h.profile_enable(False); return ms / k
cfgs = [(0, 0, 0), (8192, 0, 0), (16384, 0, 0), (32768, 0, 0), (65536, 0, 0), (131072, 0, 0), (262144, 0, 0), (16384, 4, 768), (32768, 4, 768), (16384, 2, 1024), (32768, 3, 768), (24576, 0, 0), (12288, 0, 0)]
t = {g: [] for g in cfgs}
import torch
for r in range(9):
    for g in cfgs:
        h.set_tuning(_native.TUNE_OBS_GRID, g[0]); h.set_tuning(_native.TUNE_OBS_ROWS_PER_WG, g[1]); h.set_tuning(_native.TUNE_OBS_BLOCK, g[2])
        t[g].append(timed())
        if r == 0:
            torch.cuda.synchronize(); assert torch.equal(env._t['obs'][:64], ref), g
bytes_per = 4096 * 512 * (24.0 * 512 + 24)
for g in cfgs: print("grid,rows,block", g, "median ms %.3f -> %.0f GB/s" % (statistics.median(t[g]), bytes_per / statistics.median(t[g]) / 1e6))
