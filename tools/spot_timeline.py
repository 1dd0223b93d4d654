#!/usr/bin/env python3
"""tools/spot_timeline.py [ENV_ID] [N] -- when do the workgroups of the spotlight family's one-launch step start, get their first
descriptor and finish?  Needs the measurement build lib/lab/libmemgym_spotclock.so (-DMG_LAB -DMG_LAB_SPOT_CLOCK; MEMGYM_HIP_LIB
points at it).  Prints, for the last step of a short run, microseconds from the first workgroup's start (10-ns clock)."""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "endless-memory-gym_amd"))
import memory_gym_amd  # noqa: E402
from memory_gym_amd import _native  # noqa: E402

env_id = sys.argv[1] if len(sys.argv) > 1 else "Endless-SearingSpotlights-v0"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
env = memory_gym_amd.make(env_id, num_envs=n, device=0)
env.reset(seed=torch.arange(n, dtype=torch.int64, device="cuda"))
g = torch.Generator(device="cuda").manual_seed(1)
_native.LIB.mg_lab_spot_clock.argtypes = [C.c_void_p, C.c_int, C.c_int]
for t in range(230):
    if t == 229:
        torch.cuda.synchronize()
        assert _native.LIB.mg_lab_spot_clock(None, 0, 1) == 0
    env.step(torch.randint(0, 3, (n, 2), device="cuda", generator=g, dtype=torch.int32))
torch.cuda.synchronize()
buf = np.zeros(4 * 16384, np.uint64)
assert _native.LIB.mg_lab_spot_clock(buf.ctypes.data, 16384, 0) == 0
c = buf.reshape(16384, 4).astype(np.float64)
live = c[:, 0] > 0
t0 = c[live, 0].min()
us = (c - t0) / 100.0
logic = (n + 15) // 16
idx = np.arange(16384)


def stats(name, rows, col):
    v = us[rows & (c[:, col] > 0), col]
    if len(v):
        print("%-44s n=%5d  min %6.1f  p10 %6.1f  median %6.1f  p90 %6.1f  max %6.1f us" % (name, len(v), v.min(), np.percentile(v, 10), np.median(v), np.percentile(v, 90), v.max()))


print(env_id, n, "instances;", int(live.sum()), "workgroups stamped (first 16,384 of the grid)")
stats("step WGs: start", (idx < logic) & live, 0)
stats("step WGs: end", (idx < logic) & live, 3)
stats("frame WGs: start", (idx >= logic) & live, 0)
stats("frame WGs: first descriptor there", (idx >= logic) & live, 1)
stats("frame WGs: end of last frame", (idx >= logic) & live, 3)
d = us[:, 1] - us[:, 0]
rows = (idx >= logic) & live & (c[:, 1] > 0)
print("frame WGs: wait for the first descriptor: median %.1f  p90 %.1f  max %.1f us" % (np.median(d[rows]), np.percentile(d[rows], 90), d[rows].max()))
print("launch: %.1f us from the first start to the last stamped end" % us[live][:, 3].max())
print("rescues:", env.debug_counter("one_launch_rescues"))
