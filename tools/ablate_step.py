#!/usr/bin/env python3
"""Where does the step kernel's time go?  Toggle the optional parts (reward pass, obs-table write) and time each
combination in interleaved rounds (HIP events)."""
import statistics
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from gym_d2d_amd import _native
from gym_d2d_amd.envs import VecD2DEnv
from gym_d2d_amd.envs.obs_fn import OwnLinkObsFunction


def main():
    b, c, p, r = 4096, 256, 256, 256
    env = VecD2DEnv({'num_rbs': r, 'num_cues': c, 'num_due_pairs': p, 'obs_fn': OwnLinkObsFunction}, num_envs=b)
    env.reset(seed=1)
    h = env.simulator.handle
    act = env.action_buffer()
    variants = [(rw, ob) for rw in (1, 0, 2, 3) for ob in (_native.OBS_TABLE, _native.OBS_NONE)]
    times = {v: [] for v in variants}
    for rnd in range(7):
        for v in variants:
            h.set_reward(v[0], 0.0)
            h.set_obs_mode(v[1])
            h.profile_reset(); h.profile_enable(True)
            for _ in range(10):
                h.step(act.data_ptr())
            ms, k = h.profile_read(0)
            h.profile_enable(False)
            times[v].append(ms / k * 1e3)
    for v in variants:
        print(f'reward_fn={v[0]} obs_mode={v[1]}  median {statistics.median(times[v]):7.1f} us')
    env.close()


if __name__ == '__main__':
    main()
