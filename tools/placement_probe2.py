#!/usr/bin/env python3
"""tools/placement_probe2.py -- ONE env handle, several observation buffers: steady-state raster time of real steps into
each (does the fast/slow mode belong to the observation buffer, or to something else of the handle?), next to what a
3-launch mg_render probe predicts for it.  Usage (GPU box): python tools/placement_probe2.py [ENV_ID] [N] [BUFFERS]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "endless-memory-gym_amd"))
import torch  # noqa: E402

import memory_gym_amd  # noqa: E402
from memory_gym_amd import _native  # noqa: E402

env_id = sys.argv[1] if len(sys.argv) > 1 else "MortarMayhem-Grid-v0"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
k = int(sys.argv[3]) if len(sys.argv) > 3 else 6
env = memory_gym_amd.make(env_id, num_envs=n, device=0, obs_placement="plain")
env.reset(seed=0)
g = torch.Generator(device="cuda").manual_seed(0)
hi = 4 if env.action_dim == 1 else 3
acts = [torch.randint(0, hi, (n,) if env.action_dim == 1 else (n, 2), device="cuda", generator=g, dtype=torch.int32) for _ in range(16)]
for t in range(150):  # de-synchronise the episodes first
    env.step(acts[t % 16])
bufs = [env.obs] + [torch.empty_like(env.obs) for _ in range(k - 1)]


def probe(buf):
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    _native.LIB.mg_render(env._h, buf.data_ptr(), env._stream())
    t0.record()
    for _ in range(3):
        _native.LIB.mg_render(env._h, buf.data_ptr(), env._stream())
    t1.record()
    t1.synchronize()
    return t0.elapsed_time(t1) / 3 * 1e3


for rnd in range(2):
    for i, b in enumerate(bufs):
        p = probe(b)
        env.obs = b
        env.set_profiling(1)
        for t in range(100):
            env.step(acts[t % 16])
        ms, cnt = env.get_profile(1)
        env.set_profiling(0)
        print("round %d buffer %d at 0x%x: mg_render probe %.1f us, raster in real steps %.1f us" % (rnd, i, b.data_ptr(), p, ms / cnt * 1e3), flush=True)
