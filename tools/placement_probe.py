#!/usr/bin/env python3
"""tools/placement_probe.py -- is the run-to-run bimodality of the raster kernel (e.g. MortarMayhem-Grid 231 vs 253 us in
otherwise identical runs) a property of WHERE the observation buffer lies?  One process, one env handle per trial, the
observation tensor of every trial is a fresh allocation (the earlier ones are kept alive, so the addresses differ).
Usage (GPU box): python tools/placement_probe.py [ENV_ID] [N] [TRIALS]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "endless-memory-gym_amd"))
import torch  # noqa: E402

import memory_gym_amd  # noqa: E402

env_id = sys.argv[1] if len(sys.argv) > 1 else "MortarMayhem-Grid-v0"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
trials = int(sys.argv[3]) if len(sys.argv) > 3 else 8
keep = []
g = torch.Generator(device="cuda").manual_seed(0)
for t in range(trials):
    buf = torch.empty((n, 84, 84, 3), dtype=torch.uint8, device="cuda")
    keep.append(buf)
    env = memory_gym_amd.make(env_id, num_envs=n, device=0, obs_buffer=buf)
    env.reset(seed=0)
    hi = 4 if env.action_dim == 1 else 3
    acts = [torch.randint(0, hi, (n,) if env.action_dim == 1 else (n, 2), device="cuda", generator=g, dtype=torch.int32) for _ in range(16)]
    for k in range(40):
        env.step(acts[k % 16])
    env.set_profiling(1)
    for k in range(100):
        env.step(acts[k % 16])
    ms, cnt = env.get_profile(1)
    print("trial %d: obs buffer at 0x%x (mod 2 MiB = 0x%x): raster %.1f us" % (t, buf.data_ptr(), buf.data_ptr() % (2 << 20), ms / cnt * 1e3), flush=True)
    env.close()
