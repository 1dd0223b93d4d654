#!/usr/bin/env python3
"""tools/settle_probe.py -- raster time as a function of time since the environment was created (windows of 20 steps):
how long do start-up transients (clock ramp, the driver wiping the allocator's released spacers) last?"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "endless-memory-gym_amd"))
import torch  # noqa: E402

import memory_gym_amd  # noqa: E402

env_id = sys.argv[1] if len(sys.argv) > 1 else "MortarMayhem-Grid-v0"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
t00 = time.perf_counter()
env = memory_gym_amd.make(env_id, num_envs=n, device=0)
env.reset(seed=0)
g = torch.Generator(device="cuda").manual_seed(0)
hi = 4 if env.action_dim == 1 else 3
acts = [torch.randint(0, hi, (n,) if env.action_dim == 1 else (n, 2), device="cuda", generator=g, dtype=torch.int32) for _ in range(16)]
torch.cuda.synchronize()
print("placement:", env.obs_placement_info, "setup %.2f s" % (time.perf_counter() - t00))
t0 = time.perf_counter()
out = []
for w in range(int(sys.argv[3]) if len(sys.argv) > 3 else 60):
    env.set_profiling(1)
    for t in range(20):
        env.step(acts[t % 16])
    ms, cnt = env.get_profile(1)
    out.append("%.0fms:%.0f" % ((time.perf_counter() - t0) * 1e3, ms / cnt * 1e3))
print(" ".join(out))
