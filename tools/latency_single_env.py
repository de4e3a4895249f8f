#!/usr/bin/env python3
"""Wall-clock latency of the single-env drop-in D2DEnv.step (dict in / dict out, PCIe round trips included) for the
two sizes BASELINE.md quotes for the reference (1.51 ms at 25/25/25, 81.9 ms at 256/256/256 on one Xeon core)."""
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import gym_d2d_amd


def main():
    for rbs, cues, dues in ((25, 25, 25), (256, 256, 256)):
        env = gym_d2d_amd.make('D2DEnv-v0', env_config={'num_rbs': rbs, 'num_cues': cues, 'num_due_pairs': dues})
        obs = env.reset()
        acts = {k: env.action_space['due' if k.startswith('due') else 'cue'].sample() for k in obs}
        for _ in range(3):
            env.step(acts)
        n = 30
        t0 = time.perf_counter()
        for _ in range(n):
            env.step(acts)
        dt = (time.perf_counter() - t0) / n
        t0 = time.perf_counter()
        for _ in range(n):
            env.simulator.step(env.actions)
        ds = (time.perf_counter() - t0) / n
        links = cues + dues
        print(f'{rbs}/{cues}/{dues}: D2DEnv.step {dt * 1e3:.3f} ms ({links / dt:.3g} agent-steps/s); '
              f'Simulator.step alone {ds * 1e3:.3f} ms')
        env.simulator.handle.close()


if __name__ == '__main__':
    main()
