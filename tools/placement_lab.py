#!/usr/bin/env python3
"""tools/placement_lab.py -- raster time of real MortarMayhem-Grid steps into observation buffers obtained in every way the
HIP runtime offers (hipMalloc, virtual-memory API with chosen physical chunk sizes / orders, sub-ranges of one arena),
plus the linear-fill bandwidth of single physical chunks by size.  Run on the GPU box:

    python tools/placement_lab.py [part ...]      parts: fill raster arena   (default: all)
"""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "endless-memory-gym_amd"))
import torch  # noqa: E402

import memory_gym_amd  # noqa: E402

SRC = os.path.join(ROOT, "tools", "vmm", "placement_lab.hip")
SO = os.path.join(ROOT, "tools", "vmm", "libplacement_lab.so")
if not os.path.exists(SO) or os.path.getmtime(SO) < os.path.getmtime(SRC):
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-fPIC", "-shared", "-o", SO, SRC])
LAB = C.CDLL(SO)
LAB.lab_error.restype = C.c_char_p
LAB.lab_granularity.restype = C.c_long
LAB.lab_alloc.argtypes = [C.c_size_t, C.c_int, C.c_size_t, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p)]
LAB.lab_free.argtypes = [C.c_void_p]
LAB.lab_reserve.argtypes = [C.c_size_t, C.POINTER(C.c_void_p)]
LAB.lab_create.argtypes = [C.c_size_t, C.c_int, C.POINTER(C.c_void_p)]
LAB.lab_map.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_int]
LAB.lab_unmap.argtypes = [C.c_void_p, C.c_size_t]
LAB.lab_fill_us.argtypes = [C.c_void_p, C.c_size_t, C.c_int]
LAB.lab_fill_us.restype = C.c_double

MB = 1 << 20


def alloc(nbytes, kind, chunk=0, order=0, seed=0):
    p = C.c_void_p()
    if LAB.lab_alloc(nbytes, kind, chunk, order, seed, 0, C.byref(p)) != 0:
        raise RuntimeError(LAB.lab_error().decode())
    return p.value


class _Raw:
    def __init__(self, ptr, shape):
        self.__cuda_array_interface__ = {"shape": shape, "typestr": "|u1", "data": (ptr, False), "version": 2, "strides": None}


def as_tensor(ptr, shape):
    return torch.as_tensor(_Raw(ptr, shape), device="cuda")


def part_fill():
    print("## linear-fill bandwidth of single allocations by size (best of 12; GB/s)")
    print("granularity min %d recommended %d" % (LAB.lab_granularity(0, 0), LAB.lab_granularity(0, 1)))
    for size_mb in (2, 8, 32, 128, 512, 1324):
        n = size_mb * MB
        for kind, name in ((1, "vmm-one-handle"), (0, "hipMalloc"), (2, "contiguous")):
            res = []
            ptrs = []
            for _ in range(6 if size_mb <= 128 else 3):
                try:
                    p = alloc(n, kind)
                except RuntimeError as e:
                    res.append("fail(%s)" % e)
                    continue
                ptrs.append(p)
                us = LAB.lab_fill_us(p, n, 12)
                res.append("%.0f" % (n / us / 1e3))
            for p in ptrs:
                LAB.lab_free(p)
            print("%5d MB %-15s %s" % (size_mb, name, " ".join(res)), flush=True)


def make_env(n):
    env = memory_gym_amd.make("MortarMayhem-Grid-v0", num_envs=n, device=0, obs_placement="plain")
    env.reset(seed=0)
    g = torch.Generator(device="cuda").manual_seed(0)
    acts = [torch.randint(0, 4, (n,), device="cuda", generator=g, dtype=torch.int32) for _ in range(16)]
    for t in range(150):
        env.step(acts[t % 16])
    return env, acts


def raster_us(env, acts, buf, steps=40):
    env.obs = buf
    for t in range(4):
        env.step(acts[t % 16])
    env.set_profiling(1)
    for t in range(steps):
        env.step(acts[t % 16])
    ms, cnt = env.get_profile(1)
    env.set_profiling(0)
    return ms / cnt * 1e3


def part_raster(n=65536):
    print("## raster us of real steps (MortarMayhem-Grid-v0, %d instances) into buffers by origin" % n)
    env, acts = make_env(n)
    shape = (n, 84, 84, 3)
    nbytes = n * 84 * 84 * 3
    first = env.obs
    print("torch.empty #0 (the handle's own): %.1f" % raster_us(env, acts, first), flush=True)
    specs = [("hipMalloc", 0, 0, 0), ("vmm one handle", 1, 0, 0), ("vmm 2MB chunks", 1, 2 * MB, 0), ("vmm 2MB shuffled", 1, 2 * MB, 1),
             ("vmm 2MB every-2nd", 1, 2 * MB, 2), ("vmm 32MB chunks", 1, 32 * MB, 0), ("vmm 32MB shuffled", 1, 32 * MB, 1),
             ("vmm 256MB chunks", 1, 256 * MB, 0), ("uncached", 3, 0, 0)]
    keep = []
    for rnd in range(4):
        for name, kind, chunk, order in specs:
            if kind == 3 and rnd > 0:
                continue
            try:
                p = alloc(nbytes, kind, chunk, order, seed=rnd)
            except RuntimeError as e:
                print("round %d %-18s alloc failed: %s" % (rnd, name, e), flush=True)
                continue
            keep.append(p)
            t = as_tensor(p, shape)
            us = raster_us(env, acts, t)
            fill = LAB.lab_fill_us(p, nbytes, 5)
            print("round %d %-18s raster %.1f us   (linear fill %.0f GB/s)" % (rnd, name, us, nbytes / fill / 1e3), flush=True)
    # second look at everything, in reverse order: is the mode a stable property of the buffer?
    print("-- second pass, reverse order:", " ".join("%.0f" % raster_us(env, acts, as_tensor(p, shape), 24) for p in reversed(keep)), flush=True)
    env.obs = first
    for p in keep:
        LAB.lab_free(p)
    env.close()


def part_arena(n=65536, total_gb=40):
    print("## one hipMalloc arena of %d GB, raster us into consecutive sub-ranges (MortarMayhem-Grid-v0, %d instances)" % (total_gb, n))
    env, acts = make_env(n)
    shape = (n, 84, 84, 3)
    nbytes = n * 84 * 84 * 3
    first = env.obs
    stride = (nbytes + 2 * MB - 1) // (2 * MB) * (2 * MB)
    for kind, name in ((0, "hipMalloc"), (1, "vmm one handle"), (1, "vmm 2MB chunks")):
        total = total_gb << 30
        try:
            base = alloc(total, kind, 2 * MB if name.endswith("chunks") else 0)
        except RuntimeError as e:
            print(name, "arena alloc failed:", e)
            continue
        k = total // stride
        res = [raster_us(env, acts, as_tensor(base + i * stride, shape), 24) for i in range(k)]
        print("%-16s %s" % (name, " ".join("%.0f" % r for r in res)), flush=True)
        half = [raster_us(env, acts, as_tensor(base + i * stride + stride // 2 // (2 * MB) * (2 * MB), shape), 24) for i in range(min(k - 1, 12))]
        print("%-16s (shifted by half a buffer) %s" % (name, " ".join("%.0f" % r for r in half)), flush=True)
        env.obs = first
        LAB.lab_free(base)
    env.close()


def _ck(rc):
    if rc != 0:
        raise RuntimeError(LAB.lab_error().decode())


def part_stable(n=65536):
    """Q1/Q2: is the mode a stable property of a buffer while nothing is allocated or freed?  does allocation activity move it?"""
    print("## stability: 8 hipMalloc buffers, 4 passes without allocator activity, then after 6 more allocations, then after freeing them")
    env, acts = make_env(n)
    shape, nbytes = (n, 84, 84, 3), n * 84 * 84 * 3
    first = env.obs
    bufs = [alloc(nbytes, 0) for _ in range(8)]
    print("addresses:", " ".join("0x%x" % b for b in bufs))
    for k in range(4):
        print("pass %d:" % k, " ".join("%.0f" % raster_us(env, acts, as_tensor(b, shape), 24) for b in bufs), flush=True)
    print("pass 4 (reverse order, printed in buffer order):", " ".join(reversed(["%.0f" % raster_us(env, acts, as_tensor(b, shape), 24) for b in reversed(bufs)])), flush=True)
    more = [alloc(nbytes, 0) for _ in range(6)]
    print("6 more allocated:", " ".join("%.0f" % raster_us(env, acts, as_tensor(b, shape), 24) for b in bufs), "| new ones:",
          " ".join("%.0f" % raster_us(env, acts, as_tensor(b, shape), 24) for b in more), flush=True)
    for b in more:
        LAB.lab_free(b)
    print("freed again:     ", " ".join("%.0f" % raster_us(env, acts, as_tensor(b, shape), 24) for b in bufs), flush=True)
    torch.cuda.synchronize()
    import time
    time.sleep(2.0)
    print("after 2 s idle:  ", " ".join("%.0f" % raster_us(env, acts, as_tensor(b, shape), 24) for b in bufs), flush=True)
    # longer measurement windows: does a buffer switch modes inside a long run?
    for b in bufs[:3]:
        series = []
        env.obs = as_tensor(b, shape)
        for k in range(10):
            env.set_profiling(1)
            for t in range(20):
                env.step(acts[t % 16])
            ms, cnt = env.get_profile(1)
            series.append(ms / cnt * 1e3)
        env.set_profiling(0)
        print("buffer 0x%x, ten windows of 20 steps:" % b, " ".join("%.0f" % x for x in series), flush=True)
    env.obs = first
    for b in bufs:
        LAB.lab_free(b)
    env.close()


def part_vapa(n=65536, k=5):
    """Q3: k physical handles x k virtual ranges: does the mode follow the physical memory or the virtual address?"""
    print("## raster us, rows = physical handle (hipMemCreate, one handle per buffer), columns = virtual range it is mapped at")
    env, acts = make_env(n)
    shape, nbytes = (n, 84, 84, 3), n * 84 * 84 * 3
    first = env.obs
    size = (nbytes + 2 * MB - 1) // (2 * MB) * (2 * MB)
    vas, hs = [], []
    for i in range(k):
        v, h = C.c_void_p(), C.c_void_p()
        _ck(LAB.lab_reserve(size, C.byref(v)))
        _ck(LAB.lab_create(size, 0, C.byref(h)))
        vas.append(v.value)
        hs.append(h.value)
    print("virtual ranges:", " ".join("0x%x" % v for v in vas))
    for rep in range(2):
        for i, h in enumerate(hs):
            row = []
            for v in vas:
                _ck(LAB.lab_map(v, size, h, 0))
                row.append(raster_us(env, acts, as_tensor(v, shape), 24))
                env.obs = first
                _ck(LAB.lab_unmap(v, size))
            print("rep %d handle %d: %s" % (rep, i, " ".join("%.0f" % x for x in row)), flush=True)
    env.close()


def part_scan(n=65536, total_gb=40, step_mb=256):
    print("## raster us into sub-ranges of ONE %d-GB hipMalloc arena at %d-MB steps (two sweeps)" % (total_gb, step_mb))
    env, acts = make_env(n)
    shape, nbytes = (n, 84, 84, 3), n * 84 * 84 * 3
    first = env.obs
    total = total_gb << 30
    base = alloc(total, 0)
    print("arena base 0x%x" % base)
    offs = list(range(0, total - nbytes, step_mb * MB))
    for sweep in range(2):
        res = [raster_us(env, acts, as_tensor(base + o, shape), 16) for o in offs]
        print("sweep %d: %s" % (sweep, " ".join("%.0f" % r for r in res)), flush=True)
    env.obs = first
    LAB.lab_free(base)
    env.close()


def vmm_buffer(nbytes, pieces, round_mb=2):
    """`pieces` physical handles of equal size (each a multiple of round_mb MB) mapped back to back into one reserved range"""
    unit = round_mb * MB if round_mb else 4096
    per = ((nbytes + pieces - 1) // pieces + unit - 1) // unit * unit
    v = C.c_void_p()
    _ck(LAB.lab_reserve(per * pieces, C.byref(v)))
    for i in range(pieces):
        h = C.c_void_p()
        _ck(LAB.lab_create(per, 0, C.byref(h)))
        _ck(LAB.lab_map(v.value + i * per, per, h, 0))
    return v.value


def create(nbytes):
    h = C.c_void_p()
    _ck(LAB.lab_create(nbytes, 0, C.byref(h)))
    return h.value


def map_handles(handles, per):
    v = C.c_void_p()
    _ck(LAB.lab_reserve(per * len(handles), C.byref(v)))
    for i, h in enumerate(handles):
        _ck(LAB.lab_map(v.value + i * per, per, h, 0))
    return v.value


def part_far(kind, n=65536, piece_mb=64, spacer_gb=16):
    """buffers from 64-MB physical handles: created back to back, or with spacer allocations between groups of them"""
    env, acts = make_env(n)
    shape, nbytes = (n, 84, 84, 3), n * 84 * 84 * 3
    per = piece_mb * MB
    k = (nbytes + per - 1) // per
    res = []
    for rep in range(3):
        spacers = []
        if kind == "contig":
            hs = [create(per) for _ in range(k)]
        elif kind == "reversed":
            hs = [create(per) for _ in range(k)][::-1]
        elif kind == "halves_il":  # one region, order 0, k/2, 1, k/2+1, ...
            t = [create(per) for _ in range(k)]
            hs = [t[(i // 2) + (k // 2) * (i % 2)] if i < 2 * (k // 2) else t[i] for i in range(k)]
        elif kind.startswith("split"):  # splitNN: the first NN percent from region A, spacer, the rest from region B
            cut = max(1, int(round(k * int(kind[5:]) / 100.0)))
            hs = [create(per) for _ in range(cut)]
            spacers.append(create(spacer_gb << 30))
            hs += [create(per) for _ in range(k - cut)]
        elif kind.startswith("regions"):  # regionsR: R regions (spacers between), contiguous share each
            r = int(kind[7:])
            hs = []
            for j in range(r):
                hs += [create(per) for _ in range(k * (j + 1) // r - k * j // r)]
                spacers.append(create(spacer_gb << 30))
        elif kind.startswith("rr"):  # rrR: R regions, pieces dealt round-robin
            r = int(kind[2:])
            groups = []
            for j in range(r):
                groups.append([create(per) for _ in range((k + r - 1) // r)])
                spacers.append(create(spacer_gb << 30))
            hs = [groups[i % r][i // r] for i in range(k)]
        for sp in spacers:
            LAB.lab_unmap  # (spacers were never mapped)
        p = map_handles(hs, per)
        res.append(raster_us(env, acts, as_tensor(p, shape), 24))
    print("far %-10s own %.0f | %s" % (kind, raster_us(env, acts, env.obs, 24), " ".join("%.0f" % r for r in res)), flush=True)


def release(h):
    LAB.lab_release(C.c_void_p(h))


def part_scatter(n=65536, rounds=2):
    """pieces of P MB taken from a pool `f` times as large as needed (random subset, random order; the rest is released)"""
    import random
    import time
    env, acts = make_env(n)
    shape, nbytes = (n, 84, 84, 3), n * 84 * 84 * 3
    LAB.lab_release.argtypes = [C.c_void_p]
    bufs = [("own", env.obs)]
    for rnd in range(rounds):
        for piece_mb, f in ((2, 1), (2, 4), (2, 16), (32, 1), (32, 4), (256, 1), (256, 4), (0, 1)):
            if piece_mb == 0:
                bufs.append(("r%d hipMalloc" % rnd, as_tensor(alloc(nbytes, 0), shape)))
                continue
            per = piece_mb * MB
            k = (nbytes + per - 1) // per
            pool = [create(per) for _ in range(k * f)]
            rng = random.Random(rnd * 100 + piece_mb + f)
            rng.shuffle(pool) if f > 1 else None
            for h in pool[k:]:
                release(h)
            bufs.append(("r%d %dMBx%d" % (rnd, piece_mb, f), as_tensor(map_handles(pool[:k], per), shape)))
    torch.cuda.synchronize()
    time.sleep(1.0)
    for ps in range(3):
        print("pass %d: %s" % (ps, "  ".join("%s=%.0f" % (nm, raster_us(env, acts, t, 24)) for nm, t in bufs)), flush=True)


def part_windows(n=65536, total_gb=48):
    """needs MEMGYM_HIP_LIB=.../libmemgym_lab.so (built with -DMG_LAB): the five concurrently written windows of the frame
    walk (frames [k*14336, (k+1)*14336), 303 MB each) placed independently inside one hipMalloc arena"""
    from memory_gym_amd import _native
    L = _native.LIB
    L.mg_lab_set_window_offsets.argtypes = [C.POINTER(C.c_longlong), C.c_int]
    env, acts = make_env(n)
    shape, nbytes = (n, 84, 84, 3), n * 84 * 84 * 3
    first = env.obs
    GB = 1 << 30
    W = 14336 * 21168
    base = alloc(total_gb * GB, 0)
    print("arena base 0x%x" % base)

    def run(x, offs):
        arr = (C.c_longlong * 16)(*([int(o) for o in offs] + [0] * (16 - len(offs))))
        assert L.mg_lab_set_window_offsets(arr, 16) == 0
        return raster_us(env, acts, as_tensor(base + int(x), shape), 16)

    print("natural layout at arena offset X GiB:", "  ".join("%g:%.0f" % (x, run(x * GB, [0] * 5)) for x in (1, 2, 8, 16, 24, 30, 31.25, 34, 38)), flush=True)
    for x in (2, 34):
        print("X = %d GiB, window k displaced by k*D; D =" % x, flush=True)
        for d in (0, 4096, 65536, 1 * MB, 2 * MB, 3 * MB, 5 * MB, 8 * MB, 16 * MB, 32 * MB, 48 * MB, 64 * MB, 100 * MB, 128 * MB, 209 * MB, 256 * MB, 512 * MB, GB - W, GB, 2 * GB - W, 2 * GB):
            if x * GB + 5 * W + 4 * d > total_gb * GB:
                continue
            print("   %11d (%8.1f MB): %.0f" % (d, d / MB, run(x * GB, [k * d for k in range(5)])), flush=True)
    for x in (2,):
        print("X = %d GiB, windows 0-3 natural, window 4 moved to arena offset Y GiB:" % x)
        print("  ", "  ".join("%g:%.0f" % (y, run(x * GB, [0, 0, 0, 0, y * GB - (x * GB + 4 * W)])) for y in (0, 1, 4, 8, 12, 16, 20, 24, 28, 30, 31, 32, 33, 34, 36, 40, 44)), flush=True)
        print("X = %d GiB, window 0 natural, windows 1-4 (contiguous among themselves) moved to Y GiB:" % x)
        print("  ", "  ".join("%g:%.0f" % (y, run(x * GB, [0] + [y * GB - (x * GB + W)] * 4)) for y in (4, 8, 16, 24, 30, 33, 36, 40)), flush=True)
    print("every window in its own place (GiB):")
    for places in ((1, 9, 17, 25, 35), (1, 3, 5, 7, 9), (33, 35, 37, 39, 41), (1, 2, 3, 4, 5), (1, 1.5, 2, 2.5, 3)):
        print("  ", places, "%.0f" % run(0, [pl * GB - k * W for k, pl in enumerate(places)]), flush=True)
    env.obs = first
    env.close()


def part_zones(n=65536):
    """map the zones of the whole VRAM: window 0 stays at arena offset 1 GiB, windows 1-4 are moved to Y"""
    from memory_gym_amd import _native
    L = _native.LIB
    L.mg_lab_set_window_offsets.argtypes = [C.POINTER(C.c_longlong), C.c_int]
    env, acts = make_env(n)
    shape = (n, 84, 84, 3)
    GB = 1 << 30
    W = 14336 * 21168
    base = None
    for total_gb in (272, 256, 224, 192, 128):
        try:
            base = alloc(total_gb * GB, 0)
            break
        except RuntimeError as e:
            print("arena of %d GiB: %s" % (total_gb, e))
    print("arena of %d GiB at 0x%x" % (total_gb, base), flush=True)

    def run(x, offs, steps=12):
        arr = (C.c_longlong * 16)(*([int(o) for o in offs] + [0] * (16 - len(offs))))
        assert L.mg_lab_set_window_offsets(arr, 16) == 0
        return raster_us(env, acts, as_tensor(base + int(x), shape), steps)

    x = 1
    ys = [y for y in range(4, total_gb - 2, 4)]
    print("window 0 at 1 GiB, windows 1-4 at Y:", "  ".join("%d:%.0f" % (y, run(x * GB, [0] + [y * GB - (x * GB + W)] * 4)) for y in ys), flush=True)
    print("all five windows at Y (natural layout):", "  ".join("%d:%.0f" % (y, run(y * GB, [0] * 5)) for y in ys), flush=True)
    fine = [31 + 0.125 * i for i in range(0, 12)]
    print("fine, natural layout at Y:", "  ".join("%g:%.0f" % (y, run(y * GB, [0] * 5)) for y in fine), flush=True)
    env.close()


def part_probe():
    """does a pure store kernel see the zones?  how small can the probe be?"""
    LAB.lab_fw2_us.restype = C.c_double
    LAB.lab_fw2_us.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.c_int]
    GB = 1 << 30
    base = alloc(200 * GB, 0)
    scratch = alloc(16 * MB, 0)
    ys = (3, 8, 20, 40, 60, 100, 130, 136, 150, 180, 194)
    print("fw2 (pure stores), base0 at 1 GiB, base1 at Y GiB; expected same-zone: 3 8 20 130 136 180")
    for G, nwin, lds, wrap, label in ((14336, 5, 22528, 0, "raster shape, dummy kernel between"), (14336, 5, 22528, 0, "raster shape, back to back"),
                                      (14336, 5, 0, 0, "no LDS limit (8 wg/CU)"), (1792, 36, 22528, 0, "G=1792 x 36"), (3584, 18, 22528, 0, "G=3584 x 18"),
                                      (14336, 2, 22528, 0, "two windows only"), (14336, 10, 22528, 3000, "wrap at 3000 frames (64 MB pieces), 10 windows"),
                                      (1792, 16, 22528, 0, "G=1792 x 16 (2 x 303 MB)"), (1792, 4, 22528, 0, "G=1792 x 4 (2 x 76 MB)")):
        sc = None if "back to back" in label else scratch
        res = ["%d:%.0f" % (y, LAB.lab_fw2_us(base + 1 * GB, base + y * GB, G, nwin, wrap, lds, sc, 16 * MB, 10)) for y in ys]
        print("%-55s %s" % (label, "  ".join(res)), flush=True)


def part_matrix(total_gb=256, step_gb=4):
    """two-window pure-store probe for every pair of arena positions: which pairs are slow together?"""
    import numpy as np
    LAB.lab_fw2_us.restype = C.c_double
    LAB.lab_fw2_us.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.c_int]
    GB = 1 << 30
    base = alloc(total_gb * GB, 0)
    pos = [1 + step_gb * i for i in range((total_gb - 2) // step_gb)]
    m = np.zeros((len(pos), len(pos)))
    for i, a in enumerate(pos):
        for j, b in enumerate(pos):
            if i == j:
                b = a + 0.5  # the second window right behind the first
            m[i, j] = LAB.lab_fw2_us(base + int(a * GB), base + int(b * GB), 14336, 2, 0, 22528, None, 0, 3)
    np.save(os.path.join(ROOT, "gpurun_out", "r02f", "matrix_%d.npy" % total_gb), m)
    lo, hi = np.percentile(m, 10), np.percentile(m, 90)
    print("positions (GiB):", pos)
    print("10th / 90th percentile: %.0f / %.0f us; '#' = slow pair, '.' = fast pair" % (lo, hi))
    for i in range(len(pos)):
        print("%4d " % pos[i] + "".join("#" if m[i, j] > (lo + hi) / 2 else "." for j in range(len(pos))))


def part_build(n=65536, spacer_gb=70):
    """the real construction: pieces from two (three) regions `spacer_gb` apart in allocation order, mapped alternately"""
    import time
    LAB.lab_release.argtypes = [C.c_void_p]
    LAB.lab_fw2_us.restype = C.c_double
    LAB.lab_fw2_us.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.c_int]
    env, acts = make_env(n)
    shape, nbytes = (n, 84, 84, 3), n * 84 * 84 * 3
    GB = 1 << 30
    print("own buffer: %.0f us" % raster_us(env, acts, env.obs, 24), flush=True)
    for regions in (2, 3):
        for piece_mb in (304, 152, 76, 38, 16, 4):
            per = piece_mb * MB
            k = (nbytes + per - 1) // per
            groups, spacers = [], []
            t0 = time.perf_counter()
            for r in range(regions):
                groups.append([create(per) for _ in range((k + regions - 1) // regions)])
                probe_piece = create(304 * MB)
                groups[-1].append(probe_piece)
                if r < regions - 1:
                    spacers.append(create(spacer_gb * GB))
            t1 = time.perf_counter()
            for sp in spacers:
                release(sp)
            t2 = time.perf_counter()
            probes = [g.pop() for g in groups]
            pv = [map_handles([h], 304 * MB) for h in probes]
            cls = " ".join("%.0f" % LAB.lab_fw2_us(pv[0], pv[j], 14336, 2, 0, 22528, None, 0, 3) for j in range(1, regions))
            hs = [groups[i % regions][i // regions] for i in range(k)]
            t = as_tensor(map_handles(hs, per), shape)
            r1 = raster_us(env, acts, t, 24)
            r2 = raster_us(env, acts, t, 24)
            print("%d regions, pieces of %3d MB: raster %.0f %.0f us | two-window probe region0 vs others: %s us | create %.0f ms, release spacers %.0f ms"
                  % (regions, piece_mb, r1, r2, cls, (t1 - t0) * 1e3, (t2 - t1) * 1e3), flush=True)
    print("own buffer again: %.0f us" % raster_us(env, acts, first if False else env.obs, 24))


def part_search(n=65536, step_gb=16, max_gb=200):
    """incremental search: candidate piece, probe against the first piece, spacer, next candidate ...; then build and measure"""
    import time
    LAB.lab_release.argtypes = [C.c_void_p]
    LAB.lab_fw2_us.restype = C.c_double
    LAB.lab_fw2_us.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.c_int]
    env, acts = make_env(n)
    shape, nbytes = (n, 84, 84, 3), n * 84 * 84 * 3
    GB = 1 << 30
    per = 304 * MB
    for k in range(3):
        print("own buffer: %.0f us" % raster_us(env, acts, env.obs, 24), flush=True)
    cands, spacers, log = [], [], []
    t0 = time.perf_counter()
    depth = 0
    while depth <= max_gb:
        hs = [create(per) for _ in range(3)]
        vs = [map_handles([h], per) for h in hs]
        us = LAB.lab_fw2_us(cands[0][1][0], vs[0], 14336, 2, 0, 22528, None, 0, 3) if cands else 0.0
        self_us = LAB.lab_fw2_us(vs[0], vs[1], 14336, 2, 0, 22528, None, 0, 3)
        cands.append((hs, vs))
        log.append("%d GiB: vs first %.0f, within %.0f" % (depth, us, self_us))
        try:
            spacers.append(create(step_gb * GB))
        except RuntimeError as e:
            log.append("spacer failed: %s" % e)
            break
        depth += step_gb
    t1 = time.perf_counter()
    print("search (%.1f s): %s" % (t1 - t0, " | ".join(log)), flush=True)
    for sp in spacers:
        release(sp)
    time.sleep(4.0)
    print("own buffer after the spacers were released and 4 s: %.0f us" % raster_us(env, acts, env.obs, 24), flush=True)
    # classes relative to candidate 0
    rel = [LAB.lab_fw2_us(cands[0][1][0], c[1][0], 14336, 2, 0, 22528, None, 0, 3) for c in cands[1:]]
    print("again vs first:", " ".join("%.0f" % r for r in rel))
    thr = (min(rel) + max(rel)) / 2
    other = [i + 1 for i, r in enumerate(rel) if r < thr]
    same = [0] + [i + 1 for i, r in enumerate(rel) if r >= thr]
    print("same zone as first:", same, " other:", other, flush=True)
    if other and max(rel) - min(rel) > 10:
        # third class?  probe the 'other' candidates against the first of them
        rel2 = [LAB.lab_fw2_us(cands[other[0]][1][0], cands[i][1][0], 14336, 2, 0, 22528, None, 0, 3) for i in other[1:]]
        print("others vs the first other:", " ".join("%.0f" % r for r in rel2))
        a, b = cands[same[0]], cands[other[0]]
        # every candidate holds three mapped pieces; unmap and rebuild as one buffer A B A B A
        for v in a[1] + b[1]:
            _ck(LAB.lab_unmap(v, per))
        for order, name in (([a[0][0], b[0][0], a[0][1], b[0][1], a[0][2]], "A B A B A"), ([a[0][0], a[0][1], a[0][2], b[0][0], b[0][1]], "A A A B B")):
            v = map_handles(order, per)
            t = as_tensor(v, shape)
            print("buffer %s: raster %s us" % (name, " ".join("%.0f" % raster_us(env, acts, t, 24) for _ in range(3))), flush=True)
            env.obs = env.obs  # keep
            torch.cuda.synchronize()
            for i in range(5):
                _ck(LAB.lab_unmap(v + i * per, per))
    print("own buffer at the end: %.0f us" % raster_us(env, acts, env.obs, 24), flush=True)


def part_design(n=65536, step_gb=16, max_gb=176):
    """zones found by the incremental search; piece sets allocated in each zone while the spacers are held; then buffers of
    different piece sizes / zone orders are measured (spacers still held: no background wipe traffic)"""
    import time
    LAB.lab_release.argtypes = [C.c_void_p]
    LAB.lab_fw2_us.restype = C.c_double
    LAB.lab_fw2_us.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.c_int]
    env, acts = make_env(n)
    shape, nbytes = (n, 84, 84, 3), n * 84 * 84 * 3
    GB = 1 << 30
    big = 304 * MB
    sizes = (304, 64, 16, 2)

    def piece_set():
        return {s: [create(s * MB) for _ in range((nbytes + s * MB - 1) // (s * MB))] for s in sizes}

    def probe(v0, v1):
        return LAB.lab_fw2_us(v0, v1, 14336, 2, 0, 22528, None, 0, 3)

    zones = []  # (probe va, piece set)
    spacers = []
    depth = 0
    while depth <= max_gb and len(zones) < 3:
        pv = map_handles([create(big)], big)
        ts = [probe(z[0], pv) for z in zones]
        if all(t < 107 for t in ts):
            ps = piece_set()
            pv2 = map_handles([create(big)], big)
            ts2 = [probe(z[0], pv2) for z in zones]
            print("depth %d GiB: new zone %d (probe vs earlier zones: %s; after its pieces: %s)" % (depth, len(zones), ts, ts2), flush=True)
            zones.append((pv, ps))
        spacers.append(create(step_gb * GB))
        depth += step_gb
    print("zones found:", len(zones), flush=True)
    own = raster_us(env, acts, env.obs, 24)

    def build(size_mb, pattern):
        per = size_mb * MB
        k = (nbytes + per - 1) // per
        cnt = [0] * len(zones)
        hs = []
        for i in range(k):
            z = pattern[i % len(pattern)]
            hs.append(zones[z][1][size_mb][cnt[z]])
            cnt[z] += 1
        v = map_handles(hs, per)
        t = as_tensor(v, shape)
        r = [raster_us(env, acts, t, 24) for _ in range(2)]
        env.obs = first
        torch.cuda.synchronize()
        for i in range(k):
            _ck(LAB.lab_unmap(v + i * per, per))
        return "%.0f %.0f" % tuple(r)

    first = env.obs
    print("own buffer: %.0f" % own)
    pats = [("A", [0]), ("B", [1]), ("AB", [0, 1]), ("AAB", [0, 0, 1]), ("AAAAB", [0, 0, 0, 0, 1]), ("AABB", [0, 0, 1, 1])]
    if len(zones) > 2:
        pats += [("C", [2]), ("ABC", [0, 1, 2]), ("AC", [0, 2]), ("BC", [1, 2])]
    for size_mb in sizes:
        print("pieces of %3d MB: %s" % (size_mb, " | ".join("%s: %s" % (nm, build(size_mb, p)) for nm, p in pats)), flush=True)
    print("own buffer: %.0f" % raster_us(env, acts, first, 24))
    env.close()


def part_when():
    """the store probe at the very start of a process vs after real work; constant vs varying data"""
    import time
    from memory_gym_amd import _native
    LAB.lab_fw2_us.restype = C.c_double
    LAB.lab_fw2_us.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.c_int]
    per = 304 * MB

    def pair():
        a = map_handles([create(per)], per)
        mid = [create(64 * MB) for _ in range(11)]
        b = map_handles([create(per)], per)
        return a, b

    def lib_info():
        p, info = C.c_void_p(), _native.ObsAllocInfo()
        _native.check(_native.LIB.mg_obs_alloc(0, 65536 * 21168, C.c_size_t(_native.MG_OBS_SEARCH_DEFAULT), C.byref(p), C.byref(info)), "mg_obs_alloc")
        _native.LIB.mg_obs_free(p)
        return "zones %d searched %.0f GiB same %.2f cross %.2f (%.0f ms)" % (info.zones, info.searched_bytes / 2**30, info.probe_same_tbps, info.probe_cross_tbps, info.search_ms)

    torch.cuda.init()
    a, b = pair()
    print("process start: lab fw2 on two fresh windows:", " ".join("%.0f" % LAB.lab_fw2_us(a, b, 14336, 2, 0, 22528, None, 0, 1) for _ in range(8)), flush=True)
    print("process start: mg_obs_alloc:", lib_info(), flush=True)
    x = torch.empty(1 << 30, dtype=torch.uint8, device="cuda")
    for _ in range(200):
        x.fill_(3)
    torch.cuda.synchronize()
    print("after 200 GB of fills: lab fw2 same windows:", " ".join("%.0f" % LAB.lab_fw2_us(a, b, 14336, 2, 0, 22528, None, 0, 1) for _ in range(8)), flush=True)
    a2, b2 = pair()
    print("after fills: lab fw2 on two NEW windows:", " ".join("%.0f" % LAB.lab_fw2_us(a2, b2, 14336, 2, 0, 22528, None, 0, 1) for _ in range(8)), flush=True)
    print("after fills: mg_obs_alloc:", lib_info(), flush=True)
    env, acts = make_env(65536)
    print("after env: lab fw2 first windows:", " ".join("%.0f" % LAB.lab_fw2_us(a, b, 14336, 2, 0, 22528, None, 0, 1) for _ in range(8)), flush=True)
    print("after env: mg_obs_alloc:", lib_info(), flush=True)
    print("own buffer raster: %.0f" % raster_us(env, acts, env.obs, 24))


def part_flavors():
    """do different kinds of allocation come from different ends of the VRAM?  pairwise two-window probe at process start"""
    LAB.lab_fw2_us.restype = C.c_double
    LAB.lab_fw2_us.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.c_int]
    LAB.lab_create_exportable.argtypes = [C.c_size_t, C.c_int, C.POINTER(C.c_void_p)]
    per = 304 * MB
    torch.cuda.init()
    wins = []
    for name in ("hipMalloc", "vmm", "finegrained", "vmm-exportable", "contiguous", "torch", "hipMalloc", "vmm", "managed"):
        try:
            if name == "hipMalloc":
                p = alloc(per, 0)
            elif name == "vmm":
                p = map_handles([create(per)], per)
            elif name == "vmm-exportable":
                h = C.c_void_p()
                _ck(LAB.lab_create_exportable(per, 0, C.byref(h)))
                p = map_handles([h.value], per)
            elif name == "finegrained":
                p = alloc(per, 4)
            elif name == "contiguous":
                p = alloc(per, 2)
            elif name == "torch":
                t = torch.empty(per, dtype=torch.uint8, device="cuda")
                wins.append((name, t.data_ptr(), t))
                continue
            elif name == "managed":
                t = None
                continue
            wins.append((name, p, None))
        except RuntimeError as e:
            print(name, "failed:", e)
    print("windows:", ", ".join("%s@0x%x" % (w[0], w[1]) for w in wins))
    for i, a in enumerate(wins):
        print("%-15s %s" % (a[0], " ".join("%4.0f" % (LAB.lab_fw2_us(a[1], b[1], 14336, 2, 0, 22528, None, 0, 3) if i != j else 0) for j, b in enumerate(wins))), flush=True)


def part_twokinds(n=65536):
    """buffers from ordinary and from exportable (POSIX-fd handle type) VMM pieces: all ordinary / all exportable / alternating"""
    LAB.lab_create_exportable.argtypes = [C.c_size_t, C.c_int, C.POINTER(C.c_void_p)]
    LAB.lab_fw2_us.restype = C.c_double
    LAB.lab_fw2_us.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.c_int]

    def create_x(nb):
        h = C.c_void_p()
        _ck(LAB.lab_create_exportable(nb, 0, C.byref(h)))
        return h.value

    env, acts = make_env(n)
    shape, nbytes = (n, 84, 84, 3), n * 84 * 84 * 3
    per = 64 * MB
    k = (nbytes + per - 1) // per
    own = env.obs
    print("own: %.0f" % raster_us(env, acts, own, 24))
    for rnd in range(3):
        res = []
        for name in ("ordinary", "exportable", "alternating", "alt-pairs"):
            if name == "ordinary":
                hs = [create(per) for _ in range(k)]
            elif name == "exportable":
                hs = [create_x(per) for _ in range(k)]
            elif name == "alternating":
                hs = [create(per) if i % 2 == 0 else create_x(per) for i in range(k)]
            else:
                hs = [create(per) if (i // 2) % 2 == 0 else create_x(per) for i in range(k)]
            v = map_handles(hs, per)
            t = as_tensor(v, shape)
            fill = LAB.lab_fill_us(v, nbytes, 5)
            res.append("%s: %.0f %.0f (fill %.0f GB/s)" % (name, raster_us(env, acts, t, 24), raster_us(env, acts, t, 24), nbytes / fill / 1e3))
        a = map_handles([create(304 * MB)], 304 * MB)
        b = map_handles([create_x(304 * MB)], 304 * MB)
        print("round %d: %s | probe ordinary-vs-exportable %.0f us" % (rnd, " | ".join(res), LAB.lab_fw2_us(a, b, 14336, 2, 0, 22528, None, 0, 3)), flush=True)
    print("own: %.0f" % raster_us(env, acts, own, 24))


def first_obs(env):
    return env.obs


def part_walk(pre="none"):
    """walk the allocator with spacers of several sizes; probe a fresh window against the first one at every depth"""
    from memory_gym_amd import _native
    LAB.lab_release.argtypes = [C.c_void_p]
    LAB.lab_fw2_us.restype = C.c_double
    LAB.lab_fw2_us.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.c_int]
    GB = 1 << 30
    per = 304 * MB
    torch.cuda.init()
    if pre == "env":
        env, acts = make_env(65536)
    for step_gb, keep_cands in ((16, True), (8, True), (8, False), (2, False)):
        hs = [create(per)]
        ref = map_handles(hs, per)
        spacers, log, depth = [], [], 0
        cands = []
        while depth < 160:
            spacers.append(create(step_gb * GB))
            depth += step_gb
            h = create(per)
            v = map_handles([h], per)
            us = LAB.lab_fw2_us(ref, v, 14336, 2, 0, 22528, None, 0, 3)
            log.append("%d:%.0f" % (depth, us))
            if keep_cands:
                cands.append((h, v))
            else:
                _ck(LAB.lab_unmap(v, per))
                release(h)
        print("spacers of %d GiB, candidates %s: %s" % (step_gb, "kept" if keep_cands else "released", " ".join(log)), flush=True)
        for sp in spacers:
            release(sp)
        for h, v in cands:
            _ck(LAB.lab_unmap(v, per))
            release(h)
        _ck(LAB.lab_unmap(ref, per))
        release(hs[0])
    p, info = C.c_void_p(), _native.ObsAllocInfo()
    os.environ["MEMGYM_OBS_DEBUG"] = "1"
    _native.check(_native.LIB.mg_obs_alloc(0, 65536 * 21168, C.c_size_t(160 * GB), C.byref(p), C.byref(info)), "mg_obs_alloc")
    print("library: zones %d searched %d GiB" % (info.zones, info.searched_bytes >> 30))


def part_libcheck(n=65536):
    """the library allocator's buffer under the lab's measurement (every launch bracketed) next to bench.py's (every 8th)"""
    LAB.lab_fw2_us.restype = C.c_double
    LAB.lab_fw2_us.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.c_int]
    env, acts = make_env(n)
    own = env.obs
    shape = (n, 84, 84, 3)
    print("own: %.0f" % raster_us(env, acts, own, 24))
    os.environ["MEMGYM_OBS_DEBUG"] = "1"
    for k in range(3):
        t, info = memory_gym_amd.alloc_obs_buffer(shape, torch.uint8, "cuda:0")
        print("library buffer %d: zones %d, raster %s" % (k, info["zones"], " ".join("%.0f" % raster_us(env, acts, t, 24) for _ in range(3))), flush=True)
        # stride-8 measurement like bench.py
        env.obs = t
        env.set_profiling(8)
        for s_ in range(160):
            env.step(acts[s_ % 16])
        ms, cnt = env.get_profile(1)
        env.set_profiling(0)
        print("   every 8th launch over 160 steps: %.0f us (%d launches)" % (ms / cnt * 1e3, cnt), flush=True)
        # which pieces are where: probe piece i against piece 0
        per = 304 * MB
        base = t.data_ptr()
        print("   pieces vs piece 0:", " ".join("%.0f" % LAB.lab_fw2_us(base, base + i * per, 14336, 2, 0, 22528, None, 0, 3) for i in range(1, 5)), flush=True)
    env.obs = own


def part_fresh(kind="torch", n=65536, count=5):
    """one construction per process, measured right after start (the situation bench.py is in)"""
    env, acts = make_env(n)
    shape, nbytes = (n, 84, 84, 3), n * 84 * 84 * 3
    res = [raster_us(env, acts, env.obs, 24)]
    keep = []
    for i in range(count):
        if kind == "torch":
            t = torch.empty(shape, dtype=torch.uint8, device="cuda")
        elif kind == "vmm1":
            t = as_tensor(vmm_buffer(nbytes, 1), shape)
        elif kind == "vmm1_exact":
            t = as_tensor(vmm_buffer(nbytes, 1, 0), shape)
        elif kind == "vmm2":
            t = as_tensor(vmm_buffer(nbytes, 2), shape)
        elif kind == "vmm8":
            t = as_tensor(vmm_buffer(nbytes, 8), shape)
        elif kind == "vmm21":
            t = as_tensor(vmm_buffer(nbytes, 21, 64), shape)
        keep.append(t)
        res.append(raster_us(env, acts, t, 24))
    again = [raster_us(env, acts, t, 24) for t in keep]
    print("fresh %-10s own %.0f | new: %s | again: %s" % (kind, res[0], " ".join("%.0f" % r for r in res[1:]), " ".join("%.0f" % r for r in again)), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "walk":
        part_walk(sys.argv[2])
        sys.exit(0)
    if len(sys.argv) > 2 and sys.argv[1] == "far":
        part_far(sys.argv[2])
        sys.exit(0)
    if len(sys.argv) > 2 and sys.argv[1] == "fresh":
        part_fresh(sys.argv[2])
        sys.exit(0)
    parts = sys.argv[1:] or ["fill", "raster", "arena"]
    print("device:", torch.cuda.get_device_name(0), "free GB: %.1f" % (torch.cuda.mem_get_info(0)[0] / 2**30))
    for p in parts:
        globals()["part_" + p]()
