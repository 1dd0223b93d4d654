#!/usr/bin/env python3
"""tools/store_shapes.py [N_FRAMES] -- pure store streams over a zone-balanced observation buffer (mg_obs_alloc) of N frames through
mg_store_probe: 0 linear 16-byte fill, 1 the raster's frame walk (six strided vectors per lane), 2 the frame walk with PAIRS of
adjacent vectors per lane (three rounds), 3 pairs + each wave a contiguous quarter of the frame.  Median / best of 9 launches."""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "endless-memory-gym_amd"))
from memory_gym_amd import _native  # noqa: E402
from memory_gym_amd.vec_env import alloc_obs_buffer  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
buf, info = alloc_obs_buffer((n, 84, 84, 3), torch.uint8, "cuda:0")
print("buffer: %d frames, placement %s" % (n, info))
stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
names = {0: "linear fill", 1: "frame walk, 6 strided vectors per lane (the raster's)", 2: "frame walk, 3 rounds of adjacent pairs", 3: "pairs + wave-contiguous quarters",
         4: "frame walk, line-aligned ownership", 5: "one frame per workgroup, no persistent loop",
         6: "frame walk, XCD-grouped order of frames", 7: "eight consecutive frames per workgroup, 64-byte-aligned spans"}
for rep in range(2):
    for pattern in (0, 1, 6, 7, 4, 3, 1, 6):
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(13)]
        for q in range(12):
            ev[q].record()
            _native.check(_native.LIB.mg_store_probe(C.c_void_p(buf.data_ptr()), n, pattern, stream), "mg_store_probe")
        ev[12].record()
        torch.cuda.synchronize()
        ms = sorted(ev[q].elapsed_time(ev[q + 1]) for q in range(3, 12))
        b = 21168 * n
        print("%-56s median %7.1f us %5.2f TB/s   best %7.1f us %5.2f TB/s" % (names[pattern], ms[4] * 1e3, b / ms[4] / 1e9, ms[0] * 1e3, b / ms[0] / 1e9))
