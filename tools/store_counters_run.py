#!/usr/bin/env python3
"""tools/store_counters_run.py -- the workload of tools/store_counters.sh (run under rocprofv3 --pmc ...): over ONE zone-balanced
buffer of 65,536 frames, 14 launches each of the store probes (mg_store_probe patterns 0 linear fill, 1 the raster's frame walk,
3 pairs + wave quarters, 4 line-aligned ownership, 5 one frame per workgroup), then the shipped step of MortarMayhem-Grid-v0 at
65,536 instances (mortar_step_raster_kernel) for 60 steps into the same kind of buffer.  Kernel names tell the rows apart."""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "endless-memory-gym_amd"))
import memory_gym_amd  # noqa: E402
from memory_gym_amd import _native  # noqa: E402
from memory_gym_amd.vec_env import alloc_obs_buffer  # noqa: E402

n = 65536
buf, info = alloc_obs_buffer((n, 84, 84, 3), torch.uint8, "cuda:0")
print("buffer placement:", info)
stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
for pattern in (0, 1, 3, 4, 5, 6, 7):
    for q in range(14):
        _native.check(_native.LIB.mg_store_probe(C.c_void_p(buf.data_ptr()), n, pattern, stream), "mg_store_probe")
    torch.cuda.synchronize()
env = memory_gym_amd.make("MortarMayhem-Grid-v0", num_envs=n, device=0, obs_buffer=buf)
env.reset(seed=0)
g = torch.Generator(device="cuda").manual_seed(0)
acts = [torch.randint(0, 4, (n,), device="cuda", generator=g, dtype=torch.int32) for _ in range(16)]
for t in range(260):
    env.step(acts[t % 16])
torch.cuda.synchronize()
print("done")
