#!/usr/bin/env python3
"""tools/emp_timeline.py -- when do the workgroups of Endless-MysteryPath's fused raster / service launch start and finish?
Needs a measurement build (hipcc ... -DMG_LAB_EMP_CLOCK -o lib/lab/libmemgym_empclock.so; MEMGYM_HIP_LIB points at it).
Prints, for the last step of a short run: service workgroups (end of their last served entry, end), background workgroups,
frame workgroups -- microseconds from the first workgroup's start (constant-rate clock, 10 ns)."""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "endless-memory-gym_amd"))
import memory_gym_amd  # noqa: E402
from memory_gym_amd import _native  # noqa: E402

n = 32768
env = memory_gym_amd.make("Endless-MysteryPath-v0", num_envs=n, device=0)
env.reset(seed=torch.arange(n, dtype=torch.int64, device="cuda"))
g = torch.Generator(device="cuda").manual_seed(1)
POLICY = os.environ.get("EMP_TIMELINE_POLICY", "random")  # "follower": the action read off the ground truth (eps 0.02), like tools/emp_policy_bench.py
obs, info = env.reset(seed=torch.arange(n, dtype=torch.int64, device="cuda"))
for t in range(int(os.environ.get("EMP_TIMELINE_STEPS", "260"))):
    a = torch.randint(0, 4, (n,), device="cuda", generator=g, dtype=torch.int32)
    if POLICY == "follower":
        a = torch.where(torch.rand(n, device="cuda", generator=g) < 0.02, a, info["ground_truth"].argmax(1).to(torch.int32) + 1)
    obs, rew, done, _, info = env.step(a)
torch.cuda.synchronize()
print("policy:", POLICY)
W = 14336 + 256
W = min(W, 16384)
buf = np.zeros(3 * 16384, np.uint64)
_native.LIB.mg_lab_emp_clock.argtypes = [C.c_void_p, C.c_int]
assert _native.LIB.mg_lab_emp_clock(buf.ctypes.data, 16384) == 0
c = buf.reshape(16384, 3).astype(np.float64)
live = c[:, 0] > 0
t0 = c[live, 0].min()
us = (c - t0) / 100.0
svc, bg = 64, 128  # (round 5: 64 service workgroups, one background workgroup per 256 instances)


def stats(name, rows, col):
    v = us[rows, col]
    v = v[(c[rows, col] > 0)]
    if len(v):
        print("%-34s n=%5d  min %6.1f  median %6.1f  p90 %6.1f  max %6.1f us" % (name, len(v), v.min(), np.median(v), np.percentile(v, 90), v.max()))


idx = np.arange(16384)
stats("service WGs: start", idx < svc, 0)
stats("service WGs: last entry served", idx < svc, 1)
stats("service WGs: end (frames drawn)", idx < svc, 2)
stats("background WGs: lane phase done", (idx >= svc) & (idx < svc + bg), 1)
stats("frame WGs: start", (idx >= svc) & live, 0)
stats("frame WGs: end", (idx >= svc) & live, 2)
print("launch: %.1f us from the first start to the last end" % us[live, 2].max())

# emp_step_kernel: phases per wave; every stamp follows an s_waitcnt 0, so a phase is what the wave waited for
nw = n // 64
K = 12
sb = np.zeros(K * nw, np.uint64)
_native.LIB.mg_lab_step_clock.argtypes = [C.c_void_p, C.c_int]
assert _native.LIB.mg_lab_step_clock(sb.ctypes.data, nw) == 0
sc = sb.reshape(nw, K).astype(np.float64)
s0 = sc[:, 0].min()
su = (sc - s0) / 100.0
names = {0: "wave start", 1: "state loaded, agent moved", 5: "segment records arrived", 6: "on-path test done (+ byte store)",
         7: "fall-off list / stamina flags done", 8: "direction done", 9: "reward / done / info stores issued", 10: "descriptor built",
         2: "emp_step_b returned", 3: "queue pushes done", 4: "state / descriptor stores issued"}
order = [0, 1, 5, 6, 7, 8, 9, 10, 2, 3, 4]
print("emp_step_kernel, %d waves: us from the first wave's start (stamps of waves that passed them)" % nw)
for k in order:
    v = su[:, k][sc[:, k] > 0]
    if len(v):
        print("  %-40s n=%4d  min %6.1f  median %6.1f  p90 %6.1f  max %6.1f" % (names[k], len(v), v.min(), np.median(v), np.percentile(v, 90), v.max()))
for a_, b_ in zip(order[:-1], order[1:]):
    ok = (sc[:, a_] > 0) & (sc[:, b_] > 0)
    if ok.any():
        dd = su[ok, b_] - su[ok, a_]
        print("  phase %-34s -> %-34s median %5.2f  p90 %5.2f  max %5.2f us" % (names[a_][:34], names[b_][:34], np.median(dd), np.percentile(dd, 90), dd.max()))
