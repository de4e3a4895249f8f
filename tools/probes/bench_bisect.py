#!/usr/bin/env python3
"""Which part of bench.py's process history puts the config-2 session into the all-slow placement state (every one of 15
candidate obs blocks 14.9 us, where a fresh process finds 13.2)?  One variant per process."""
import gc
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
import torch

import bench

variant = sys.argv[1]
args = bench.parse_args(['--no-cpu-baseline'])
dev = torch.device('cuda', 0)
torch.cuda.set_device(0)


def fence():
    torch.cuda.synchronize(dev)


def default_session(tag):
    s = bench.Session(torch, args, 'default', 'linear', dev, 0, 0, 500, 500, action_pool=64)
    t = s.timed(fence)
    print(json.dumps({'variant': variant, 'after': tag, 'us_per_step': round(t['dt'] / 500 * 1e6, 2), 'placement': s.env.placement,
                      'torch_reserved_GB': round(torch.cuda.memory_reserved() / 1e9, 2)}), flush=True)
    s.close()


if variant == 'single_env_first':
    bench.single_env_latency(25, 25, 25, 0)
    bench.single_env_latency(256, 256, 256, 0)
    default_session('the two single-env latency runs')
elif variant == 'headline_short':
    sess = bench.Session(torch, args, 'stress', 'linear', dev, 0, 0, 10, 2)
    sess.timed(fence)
    sess.close()
    default_session('a 12-step headline session, closed (object alive)')
elif variant == 'headline_full':
    sess = bench.Session(torch, args, 'stress', 'linear', dev, 0, 0, 100, 10)
    sess.timed(fence)
    sess.close()
    default_session('a 110-step headline session, closed (object alive)')
    del sess
    gc.collect()
    default_session('the session object deleted')
    torch.cuda.empty_cache()
    default_session('torch.cuda.empty_cache()')
elif variant == 'headline_core':
    sess = bench.Session(torch, args, 'stress', 'linear', dev, 0, 0, 100, 10)
    sess.timed(fence)
    bench.core_mode(torch, sess, dev, args, fence, False, 1, None)
    sess.close()
    default_session('headline + core_mode, closed')
    del sess
    gc.collect()
    torch.cuda.empty_cache()
    default_session('deleted + empty_cache')
else:
    default_session('nothing')
