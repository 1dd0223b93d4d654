#!/usr/bin/env python3
"""tools/placement_probe4.py -- the converse of placement_probe2: TWO observation buffers (fixed), several env handles
(each with its own state / descriptor / atlas allocations, earlier handles kept alive): does the raster's mode follow the
handle's own arrays as well?  Usage (GPU box): python tools/placement_probe4.py [ENV_ID] [N] [HANDLES]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "endless-memory-gym_amd"))
import torch  # noqa: E402

import memory_gym_amd  # noqa: E402

env_id = sys.argv[1] if len(sys.argv) > 1 else "MortarMayhem-Grid-v0"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
handles = int(sys.argv[3]) if len(sys.argv) > 3 else 5
bufs = [torch.empty((n, 84, 84, 3), dtype=torch.uint8, device="cuda") for _ in range(6)]
g = torch.Generator(device="cuda").manual_seed(0)
keep = []
for h in range(handles):
    env = memory_gym_amd.make(env_id, num_envs=n, device=0, obs_placement="plain", obs_buffer=bufs[0])
    keep.append(env)
    env.reset(seed=0)
    hi = 4 if env.action_dim == 1 else 3
    acts = [torch.randint(0, hi, (n,) if env.action_dim == 1 else (n, 2), device="cuda", generator=g, dtype=torch.int32) for _ in range(16)]
    for t in range(150):
        env.step(acts[t % 16])
    out = []
    for b in bufs:
        env.obs = b
        for t in range(6):
            env.step(acts[t % 16])
        env.set_profiling(1)
        for t in range(40):
            env.step(acts[t % 16])
        ms, cnt = env.get_profile(1)
        env.set_profiling(0)
        out.append(ms / cnt * 1e3)
    print("handle %d: raster us into buffers 0..5: %s" % (h, " ".join("%.1f" % x for x in out)), flush=True)
