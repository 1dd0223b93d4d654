"""tools/path_lab.py -- where the time of Endless-MysteryPath's lane-per-job path generator goes.

Needs a measurement build of the library (clock64() around the phases of lane_path, counters read back through
mg_lab_path_stats):
    cd endless-memory-gym_amd && hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -DMG_LAB_PATH \
        -o lib/libmemgym_lab.so csrc/*.hip
    MEMGYM_HIP_LIB=endless-memory-gym_amd/lib/libmemgym_lab.so MEMGYM_EMP_FUSE=0 MEMGYM_EMP_LANES=1 python tools/path_lab.py
"""
import ctypes
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "endless-memory-gym_amd"))
import memory_gym_amd as mg  # noqa: E402
from memory_gym_amd import _native  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
    lib = _native.LIB
    lib.mg_lab_path_stats.argtypes = [ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int]
    env = mg.make("Endless-MysteryPath-v0", num_envs=n, device="cuda:0")
    env.reset(seed=list(range(n)))
    out = (ctypes.c_ulonglong * 16)()
    torch.cuda.synchronize()
    lib.mg_lab_path_stats(out, 1)
    g = torch.Generator(device="cuda:0").manual_seed(1)
    acts = [torch.randint(0, 4, (n,), device="cuda:0", dtype=torch.int32, generator=g) for _ in range(16)]
    for k in range(100):
        env.step(acts[k % 16])
    torch.cuda.synchronize()
    lib.mg_lab_path_stats(out, 1)
    t0 = time.perf_counter()
    for k in range(steps):
        env.step(acts[k % 16])
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    lib.mg_lab_path_stats(out, 1)
    v = list(out)
    calls = max(v[3], 1)
    print("step %.1f us" % (dt * 1e6))
    print("per wave call of lane_path: walls %.0f ticks, whole %.0f ticks, expansions (lane 0 loop trips) %.1f" %
          (v[0] / calls, v[1] / calls, v[2] / calls))
    print("waves with jobs per step %.1f, path calls per step %.1f, kernel ticks per busy wave %.0f" %
          (v[5] / steps, v[3] / steps, v[4] / max(v[5], 1)))


if __name__ == "__main__":
    main()
