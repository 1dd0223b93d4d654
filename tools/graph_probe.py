#!/usr/bin/env python3
"""tools/graph_probe.py -- capture K consecutive mg_step calls (fixed action buffers) in a HIP graph and compare the
replay rate with eager launches for small batches, where the two launches per step are host-bound.
Usage (GPU box): python tools/graph_probe.py ENV_ID N_ENVS [K]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "endless-memory-gym_amd"))
import torch  # noqa: E402

import memory_gym_amd  # noqa: E402

env_id, n = sys.argv[1], int(sys.argv[2])
K = int(sys.argv[3]) if len(sys.argv) > 3 else 16
env = memory_gym_amd.make(env_id, num_envs=n, device=0)
env.reset(seed=0)
g = torch.Generator(device="cuda").manual_seed(0)
hi = 4 if env.action_dim == 1 else 3
shape = (n,) if env.action_dim == 1 else (n, 2)
acts = [torch.randint(0, hi, shape, device="cuda", generator=g, dtype=torch.int32) for _ in range(K)]
for a in acts:
    env.step(a)
torch.cuda.synchronize()
t0 = time.perf_counter()
R = 50
for _ in range(R):
    for a in acts:
        env.step(a)
torch.cuda.synchronize()
eager = n * K * R / (time.perf_counter() - t0)

side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for a in acts:
        env.step(a)
torch.cuda.current_stream().wait_stream(side)
graph = torch.cuda.CUDAGraph()
with torch.cuda.graph(graph):
    for a in acts:
        env.step(a)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(R):
    graph.replay()
torch.cuda.synchronize()
replay = n * K * R / (time.perf_counter() - t0)
env.check_errors()
print("%s n=%d: eager %.2f M steps/s, graph replay of %d steps %.2f M steps/s (x%.2f)" % (env_id, n, eager / 1e6, K, replay / 1e6, replay / eager))
