#!/usr/bin/env python3
"""tools/groups_probe.py -- env-steps/s of mg_step with 1, 2, 4, 8 instance groups (include/memgym.h: mg_set_groups), same
process, same box, BASELINE sizes; every configuration twice, interleaved.  Prints one table row per (workload, groups)."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "endless-memory-gym_amd"))
import memory_gym_amd  # noqa: E402

WORK = [("MortarMayhem-Grid-v0", 65536), ("MysteryPath-v0", 32768), ("Endless-SearingSpotlights-v0", 16384), ("Endless-MortarMayhem-v0", 32768),
        ("SearingSpotlights-v0", 16384), ("MysteryPath-Grid-v0", 32768), ("Endless-MysteryPath-v0", 32768)]


def rate(env_id, n, groups, steps=300, settle=200):
    env = memory_gym_amd.make(env_id, num_envs=n, device=0, groups=groups)
    env.reset(seed=torch.arange(n, dtype=torch.int64, device="cuda"))
    g = torch.Generator(device="cuda").manual_seed(5)
    shape, hi = ((n,), 4) if env.action_dim == 1 else ((n, 2), 3)
    acts = [torch.randint(0, hi, shape, device="cuda", generator=g, dtype=torch.int32) for _ in range(32)]
    for k in range(settle):
        env.step(acts[k % 32])
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    e0.record()
    for k in range(steps):
        env.step(acts[k % 32])
    e1.record()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 1e3 / steps  # host clock: with MEMGYM_GROUPS_LAB=1 the blocks' streams are not joined into torch's
    zones = (env.obs_placement_info or {}).get("zones")
    env.close()
    return n / ms / 1e3, ms * 1e3, zones


GROUPS = tuple(int(x) for x in os.environ.get("PROBE_GROUPS", "1,2,4,8").split(","))


def main():
    only = sys.argv[1:]
    print("| workload | instances | groups | M env-steps/s (two runs) | us per step | zones |")
    print("|---|---:|---:|---|---|---|")
    for env_id, n in WORK:
        if only and env_id not in only:
            continue
        res = {}
        for rep in range(2):
            for groups in GROUPS:
                res.setdefault(groups, []).append(rate(env_id, n, groups))
        for groups in GROUPS:
            r = res[groups]
            print("| %s | %d | %d | %s | %s | %s |" % (env_id, n, groups, " / ".join("%.1f" % x[0] for x in r), " / ".join("%.1f" % x[1] for x in r),
                                                     "/".join(str(x[2]) for x in r)), flush=True)


if __name__ == "__main__":
    main()
