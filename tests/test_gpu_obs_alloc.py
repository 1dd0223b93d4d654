"""GPU (-m gpu): mg_obs_alloc (zone-balanced observation buffers, include/memgym.h) under conditions a trainer creates:
most of the VRAM held by somebody else, buffers created and destroyed over and over, a runtime without the virtual-memory
API.  In every case the environment must work (frames equal the plain-allocation run), the call must return in bounded
time, and what could not be had must be reported (info.zones), not guessed."""
import ctypes as C
import os
import subprocess
import sys
import time

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LAB_LIB = os.path.join(ROOT, "endless-memory-gym_amd", "lib", "lab", "libmemgym_hip_lab.so")  # the hook exists in the -DMG_LAB build only (csrc/mg_lab.hpp)


def _stats():
    from memory_gym_amd import _native
    a, b, c = C.c_size_t(), C.c_size_t(), C.c_size_t()
    assert _native.LIB.mg_obs_debug_stats(C.byref(a), C.byref(b), C.byref(c)) == 0
    return a.value, b.value, c.value


def _frames(env, steps=5):
    n = env.num_envs
    env.reset(seed=torch.arange(n, dtype=torch.int64, device="cuda"))
    g = torch.Generator(device="cuda").manual_seed(1)
    for _ in range(steps):
        obs, *_ = env.step(torch.randint(0, 4, (n,), device="cuda", generator=g, dtype=torch.int32))
    return obs


def test_most_of_the_vram_held_by_a_model():
    """A 200-GiB tensor is allocated first (a model, a replay buffer): the search has little room, must stay within its
    budget and time bound, and the environment produces the same frames as with a plain buffer."""
    import memory_gym_amd

    free, total = torch.cuda.mem_get_info()
    hold_gib = min(200, int(free / (1 << 30)) - 24)
    if hold_gib < 32:
        pytest.skip("not enough free VRAM on this box")
    hold = torch.empty(hold_gib << 30, dtype=torch.uint8, device="cuda")
    n = 32768  # 694 MB of observations: three pieces
    t0 = time.perf_counter()
    env = memory_gym_amd.make("MortarMayhem-Grid-v0", num_envs=n, device=0)
    dt = time.perf_counter() - t0
    info = env.obs_placement_info
    assert dt < 8.0, "make() took %.1f s next to a %d-GiB tensor" % (dt, hold_gib)
    assert info is None or (info["zones"] in (0, 1, 2, 3) and info["searched_bytes"] <= 129 << 30)
    ref = memory_gym_amd.make("MortarMayhem-Grid-v0", num_envs=n, device=0, obs_placement="plain")
    assert torch.equal(_frames(env), _frames(ref))
    env.close()
    ref.close()
    del hold, env, ref
    torch.cuda.empty_cache()


def test_hundred_create_destroy_cycles_do_not_leak():
    import gc

    import memory_gym_amd

    n = 32768
    torch.cuda.synchronize()
    free0, _ = torch.cuda.mem_get_info()
    live0, _, va0 = _stats()
    worst = 0.0
    for k in range(100):
        t0 = time.perf_counter()
        env = memory_gym_amd.make("MortarMayhem-Grid-v0", num_envs=n, device=0)
        worst = max(worst, time.perf_counter() - t0)
        if k % 25 == 0:
            _frames(env, 2)
        env.close()
        del env
        gc.collect()
    torch.cuda.synchronize()
    free1, _ = torch.cuda.mem_get_info()
    live1, pooled, va1 = _stats()
    assert live1 == live0, "observation buffers still alive: %d" % (live1 - live0)
    assert pooled <= 10
    # the pool keeps at most ten 304-MiB pieces; everything else went back to the driver
    assert free0 - free1 <= (11 * 304 << 20) + (64 << 20), "VRAM not returned: %.2f GiB" % ((free0 - free1) / 2 ** 30)
    # address space is never handed back (stale translations on ROCm 7.2): bounded use, far from the 128-TiB space
    assert va1 - va0 < 1 << 40, "%.1f GiB of address space for 100 buffers" % ((va1 - va0) / 2 ** 30)
    assert worst < 8.0  # the search is bounded (1.5 s by default: mg_obs_set_search_ms); the rest is mg_create


WORKER = r'''
import os, sys
sys.path.insert(0, os.path.join(%(root)r, "endless-memory-gym_amd"))
import torch, memory_gym_amd
n = 32768
env = memory_gym_amd.make("MortarMayhem-Grid-v0", num_envs=n, device=0)
info = env.obs_placement_info
assert info is None or info["zones"] <= 1, info   # no virtual-memory API: the plain path, and it says so
env.reset(seed=torch.arange(n, dtype=torch.int64, device="cuda"))
obs, *_ = env.step(torch.zeros(n, dtype=torch.int32, device="cuda"))
torch.cuda.synchronize()
assert int(obs.sum()) > 0
print("NO_VMM_OK")
'''


def test_runtime_without_the_virtual_memory_api_gets_a_plain_buffer():
    out = subprocess.run([sys.executable, "-c", WORKER % {"root": ROOT}], env=dict(os.environ, MEMGYM_OBS_NO_VMM="1", MEMGYM_HIP_LIB=LAB_LIB), capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "NO_VMM_OK" in out.stdout, out.stderr[-2000:]


def test_a_balanced_buffer_is_never_exported_through_hip_ipc():
    """mg_obs_alloc memory (hipMemCreate pieces in a reserved range) has no IPC handle; PeerObsBuffer refuses it by name instead
    of failing inside torch's reduce_tensor (multi-GPU readiness)."""
    import torch
    from memory_gym_amd.dist import PeerObsBuffer
    from memory_gym_amd.vec_env import alloc_obs_buffer, is_balanced_buffer

    t, info = alloc_obs_buffer((20000, 84, 84, 3), torch.uint8, "cuda:0")
    plain = torch.empty((16, 84, 84, 3), dtype=torch.uint8, device="cuda:0")
    assert not is_balanced_buffer(plain)
    PeerObsBuffer._check_exportable(plain)
    if info["pieces"] > 0:
        assert is_balanced_buffer(t) and is_balanced_buffer(t[100:200])
        with pytest.raises(ValueError):
            PeerObsBuffer._check_exportable(t)
    del t


BOUND_WORKER = r'''
import os, sys, json
sys.path.insert(0, os.path.join(%(root)r, "endless-memory-gym_amd"))
import torch
from memory_gym_amd.vec_env import alloc_obs_buffer
import time, ctypes
from memory_gym_amd import _native
warm = torch.empty(4 * 21168, dtype=torch.uint8, device="cuda:0")  # the library's kernels are loaded by their first launch (seconds on a box
_native.LIB.mg_store_probe(ctypes.c_void_p(warm.data_ptr()), 4, 1, None)  # with slow storage): not the search's time
torch.cuda.synchronize()
t0 = time.perf_counter()
plain = torch.empty((65536, 84, 84, 3), dtype=torch.uint8, device="cuda:0")  # what the driver takes to hand out 1.4 GB at all (it wipes dirtied memory)
torch.cuda.synchronize()
plain_ms = (time.perf_counter() - t0) * 1e3
del plain
torch.cuda.empty_cache()
out = [plain_ms]
for k in range(6):
    t, info = alloc_obs_buffer((65536, 84, 84, 3), torch.uint8, "cuda:0")
    t[::4096].fill_(7)
    torch.cuda.synchronize()
    out.append((info["search_ms"], info["zones"], info["searched_bytes"]))
    if k %% 2:
        del t  # (every second buffer stays alive: the next search cannot be served from the pool alone)
print("BOUND " + json.dumps(out))
'''


@pytest.mark.parametrize("bound_ms", [50, 400])
def test_the_search_time_bound_is_a_bound(bound_ms):
    """info.search_ms stays within 1.5 x the bound (MEMGYM_OBS_SEARCH_MS -> mg_obs_set_search_ms) -- the walk reads the
    clock after every handle it creates and projects what handing everything back will cost; it used to look at the head of its
    loop only (a 1.5-s bound, 5 s measured under rocprofv3).  A search that runs out of time ends with a usable buffer."""
    out = subprocess.run([sys.executable, "-c", BOUND_WORKER % {"root": ROOT}], env=dict(os.environ, MEMGYM_OBS_SEARCH_MS=str(bound_ms)),
                         capture_output=True, text=True, timeout=600)
    line = [ln for ln in out.stdout.splitlines() if ln.startswith("BOUND ")]
    assert out.returncode == 0 and line, out.stderr[-2000:]
    import json
    plain_ms, *rows = json.loads(line[-1][6:])
    # (the bound is the WALK's.  What no bound can take away is handing out the buffer itself: its five 304-MiB pieces -- on memory an
    # earlier process dirtied the driver wipes what it hands out, ~27 ms per GiB -- their probes, and the plain allocation a search
    # that ran out of time ends with: 100-115 ms on such a box with nothing walked at all, hence the fixed allowance)
    for ms, zones, walked in rows:
        assert ms <= 1.5 * bound_ms + 100.0 + 2.0 * plain_ms, "search_ms %.0f with a bound of %d ms (zones %d, %.1f GiB walked; a plain allocation: %.0f ms)" % (
            ms, bound_ms, zones, walked / 2 ** 30, plain_ms)
