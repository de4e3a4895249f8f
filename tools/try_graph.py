#!/usr/bin/env python3
"""Can d2d_step be captured into a HIP graph (torch.cuda.graph) and replayed?  Steady-state d2d_step makes no
allocation and no synchronisation, so it should be; this measures what a replayed 10-step episode costs on the
launch-bound default workload."""
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch

from gym_d2d_amd.envs import VecD2DEnv


def main():
    b, c, p, r = map(int, sys.argv[1:5]) if len(sys.argv) > 1 else (1024, 25, 25, 25)
    n = c + p
    env = VecD2DEnv({'num_rbs': r, 'num_cues': c, 'num_due_pairs': p}, num_envs=b)
    env.reset(seed=1)
    h = env.simulator.handle
    dev = env.device
    acts = torch.randint(0, r * 21, (10, b, n), device=dev, dtype=torch.int32)
    # eager reference
    for k in range(10):
        h.step(acts[k].data_ptr())
    torch.cuda.synchronize()
    ref = env._t['sinr_db'].clone()
    side = torch.cuda.Stream(device=dev)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        env._follow_torch_stream()
        h.step(acts[0].data_ptr())               # warm-up on the capture stream
        side.synchronize()
        with torch.cuda.graph(g, stream=side):
            for k in range(10):
                h.step(acts[k].data_ptr())
    torch.cuda.synchronize()
    env._t['sinr_db'].zero_()
    g.replay()
    torch.cuda.synchronize()
    print('graph replay reproduces eager results:', torch.equal(env._t['sinr_db'], ref))
    for name, fn in (('graph replay (10 steps)', g.replay),):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(50):
            fn()
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 500
        print(f'{name}: {dt * 1e6:.1f} us per step -> {b * n / dt / 1e9:.2f} G agent-steps/s')
    with torch.cuda.stream(side):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(50):
            for k in range(10):
                h.step(acts[k].data_ptr())
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 500
    print(f'eager: {dt * 1e6:.1f} us per step -> {b * n / dt / 1e9:.2f} G agent-steps/s')
    env.close()


if __name__ == '__main__':
    main()
