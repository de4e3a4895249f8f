#!/usr/bin/env python3
"""Does the PHYSICAL make-up of the 61 MB obs block decide BASELINE config 2's speed class?  The real env, its obs block bound to
memory from tools/probes/vmm_alloc.hip: hipMalloc / one physical handle / 2 MiB chunks in order / shuffled (three seeds) / every
other created chunk skipped; eight allocations of each kind, all held."""
import ctypes as C
import json
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import torch

from gym_d2d_amd import _native
from gym_d2d_amd.envs import VecD2DEnv

lib = C.CDLL('/tmp/libvmm_alloc.so')
lib.vmm_alloc.argtypes = [C.c_size_t, C.c_int, C.c_ulonglong, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]
env = VecD2DEnv({'num_rbs': 25, 'num_cues': 25, 'num_due_pairs': 25}, num_envs=1024, cue_actions='traffic', placement_trials=0)
env.reset(seed=1)
h = env.simulator.handle
acts = torch.randint(0, 25 * 21, (8, 1024, 25), device=env.device, dtype=torch.int32)
nbytes = 1024 * 50 * 300 * 4


def steady(steps=800):
    for k in range(100):
        h.step(acts[k % 8].data_ptr())
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(steps):
        h.step(acts[k % 8].data_ptr())
    torch.cuda.synchronize()
    return round((time.perf_counter() - t0) / steps * 1e6, 2)


for k in range(3000):
    h.step(acts[k % 8].data_ptr())
print(json.dumps({'backing': 'torch.empty (as created)', 'us_per_step': [steady()]}), flush=True)
NAMES = {0: 'hipMalloc', 1: 'VMM, one physical handle', 2: 'VMM, 2 MiB chunks in creation order', 3: 'VMM, 2 MiB chunks shuffled',
         4: 'VMM, every other created 2 MiB chunk (neighbours not adjacent)'}
for rnd in range(2):
    for mode in (0, 1, 2, 3, 4):
        res = []
        for k in range(8):
            ptr, gran = C.c_void_p(), C.c_size_t()
            rc = lib.vmm_alloc(nbytes, mode, rnd * 100 + k, C.byref(ptr), C.byref(gran))
            if rc or not ptr.value:
                res.append(None)
                continue
            h.bind_buffer(_native.BUF_OBS, ptr.value, nbytes)
            res.append(steady())
        print(json.dumps({'round': rnd, 'backing': NAMES[mode], 'granularity': gran.value, 'us_per_step': res}), flush=True)
env.close()
