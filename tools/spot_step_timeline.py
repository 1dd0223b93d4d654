#!/usr/bin/env python3
"""Where the spotlight family's step kernel spends its time, wave by wave (lab build with -DMG_LAB_SPOT_CLOCK:
tools/build_variant.sh clock mg_spot.hip -DMG_LAB_SPOT_CLOCK; MEMGYM_HIP_LIB=.../libmemgym_clock.so).
Eight s_memtime stamps per wave (entry, state in, spawn done, hits done, coin / done logic, before the reset / descriptor part,
before the final stores, stores landed) + flags (any instance of the wave spawned / re-sampled its coin / reset) + the 100-MHz
clock at entry and exit.  usage: spot_step_timeline.py [env_id] [n] [steps]"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "endless-memory-gym_amd"))
import torch  # noqa: E402

import memory_gym_amd  # noqa: E402
from memory_gym_amd import _native  # noqa: E402

env_id = sys.argv[1] if len(sys.argv) > 1 else "Endless-SearingSpotlights-v0"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
L = _native.LIB
L.mg_lab_spot_clock.argtypes = [C.c_void_p, C.c_int, C.c_int]
env = memory_gym_amd.make(env_id, num_envs=n, device=0)
env.reset(seed=np.arange(n, dtype=np.int64))
g = torch.Generator(device="cuda").manual_seed(3)
acts = [torch.randint(0, 3, (n, 2), generator=g, device="cuda", dtype=torch.int32) for _ in range(120 + steps)]
for t in range(120):
    env.step(acts[t])
torch.cuda.synchronize()
waves = min(n // 4, 65536)
names = ["state in", "spawn", "slots+hits", "coin/done", "bookkeeping", "reset|desc", "stores"]
for t in range(steps):
    assert L.mg_lab_spot_clock(None, 0, 1) == 0
    env.step(acts[120 + t])
    torch.cuda.synchronize()
    buf = np.zeros((waves, 10), dtype=np.uint64)
    assert L.mg_lab_spot_clock(buf.ctypes.data, waves, 0) == 0
    clk = buf[:, :8].astype(np.int64)
    flags = buf[:, 8].astype(np.int64)
    w0 = (buf[:, 9] & np.uint64(0xFFFFFFFF)).astype(np.int64)
    w1 = (buf[:, 9] >> np.uint64(32)).astype(np.int64)
    w1 = np.where(w1 < w0, w1 + (1 << 32), w1)
    wall = (w1 - w0) * 10.0  # ns
    cyc = clk[:, 7] - clk[:, 0]
    ns_per_cycle = np.median(wall[cyc > 0] / cyc[cyc > 0])
    seg = np.diff(clk, axis=1) * ns_per_cycle / 1e3  # us
    start = (w0 - w0.min()) * 0.01  # us
    end = (w1 - w0.min()) * 0.01
    print("step %d: %d waves, %.3f ns per s_memtime tick; first wave starts 0, last starts %.1f us, kernel ends %.1f us" % (t, waves, ns_per_cycle, start.max(), end.max()))
    for f in sorted(set(flags.tolist())):
        m = flags == f
        label = "+".join(x for b, x in ((1, "spawn"), (2, "coin"), (4, "reset")) if f & b) or "plain"
        tot = seg[m].sum(1)
        print("  %-18s %5d waves  total %5.1f / %5.1f / %5.1f us (median / p90 / max)   " % (label, m.sum(), np.median(tot), np.percentile(tot, 90), tot.max())
              + "  ".join("%s %.1f" % (names[k], np.median(seg[m][:, k])) for k in range(7)))
    late = np.argsort(end)[-5:]
    print("  last five waves to end: " + "; ".join("wave %d flags %d start %.1f end %.1f" % (w, flags[w], start[w], end[w]) for w in late))
env.close()
