#!/usr/bin/env python3
"""tools/logic_probe.py -- time the logic / raster kernels of one workload under reset options (HIP events every step).
Usage (GPU box): python tools/logic_probe.py ENV_ID N_ENVS '{"agent_health": 100000}' [steps]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "endless-memory-gym_amd"))
import torch  # noqa: E402

import memory_gym_amd  # noqa: E402

env_id, n = sys.argv[1], int(sys.argv[2])
options = json.loads(sys.argv[3]) if len(sys.argv) > 3 and sys.argv[3] else None
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 200
env = memory_gym_amd.make(env_id, num_envs=n, device=0)
env.reset(seed=0, options=options)
g = torch.Generator(device="cuda").manual_seed(0)
hi = 4 if env.action_dim == 1 else 3
shape = (n,) if env.action_dim == 1 else (n, 2)
acts = [torch.randint(0, hi, shape, device="cuda", generator=g, dtype=torch.int32) for _ in range(32)]
for t in range(50):
    env.step(acts[t % 32])
env.set_profiling(1)
dones = 0
for t in range(steps):
    _, _, d, _, _ = env.step(acts[t % 32])
    dones += int(d.sum().item())
lm, ln = env.get_profile(0)
rm, rn = env.get_profile(1)
print(json.dumps(dict(env_id=env_id, n=n, options=options, logic_us=lm / ln * 1e3 if ln else None, raster_us=rm / rn * 1e3,
                      done_rate_per_step=dones / steps / n)))
