#!/usr/bin/env python3
"""tools/overlap_probe.py -- double-buffered sampling: two handles of N/2 instances stepped on two HIP streams (the logic
kernel of one group runs under the raster kernel of the other) vs one handle of N instances on one stream.
Usage (GPU box): python tools/overlap_probe.py ENV_ID N"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "endless-memory-gym_amd"))
import torch  # noqa: E402

import memory_gym_amd  # noqa: E402

env_id, n = sys.argv[1], int(sys.argv[2])
K = 300


def acts_for(env, m, seed):
    g = torch.Generator(device="cuda").manual_seed(seed)
    hi = 4 if env.action_dim == 1 else 3
    shape = (m,) if env.action_dim == 1 else (m, 2)
    return [torch.randint(0, hi, shape, device="cuda", generator=g, dtype=torch.int32) for _ in range(32)]


one = memory_gym_amd.make(env_id, num_envs=n, device=0)
one.reset(seed=0)
a1 = acts_for(one, n, 0)
for k in range(30):
    one.step(a1[k % 32])
torch.cuda.synchronize()
t0 = time.perf_counter()
for k in range(K):
    one.step(a1[k % 32])
torch.cuda.synchronize()
single = n * K / (time.perf_counter() - t0)
one.close()

halves = [memory_gym_amd.make(env_id, num_envs=n // 2, device=0) for _ in range(2)]
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
acts = []
for h, e in enumerate(halves):
    e.reset(seed=h * (n // 2))
    acts.append(acts_for(e, n // 2, 1 + h))
torch.cuda.synchronize()
for k in range(30):
    for e, s, a in zip(halves, streams, acts):
        with torch.cuda.stream(s):
            e.step(a[k % 32])
torch.cuda.synchronize()
t0 = time.perf_counter()
for k in range(K):
    for e, s, a in zip(halves, streams, acts):
        with torch.cuda.stream(s):
            e.step(a[k % 32])
torch.cuda.synchronize()
dual = n * K / (time.perf_counter() - t0)
print("%s n=%d: one stream %.1f M steps/s, two half-batches on two streams %.1f M steps/s (x%.2f)" % (env_id, n, single / 1e6, dual / 1e6, dual / single))
