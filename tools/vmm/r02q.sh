# does the release of walked spacers disturb the raster afterwards, and for how long?
python - <<'PY'
import os, sys, time, ctypes as C
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "endless-memory-gym_amd"))
import torch, memory_gym_amd
from memory_gym_amd import _native
n = 32768
env = memory_gym_amd.make("MysteryPath-v0", num_envs=n, device=0)
print("placement", env.obs_placement_info)
env.reset(seed=0)
g = torch.Generator(device="cuda").manual_seed(0)
acts = [torch.randint(0, 3, (n, 2), device="cuda", generator=g, dtype=torch.int32) for _ in range(16)]
def series(k):
    out = []
    for w in range(k):
        env.set_profiling(1)
        for t in range(20): env.step(acts[t % 16])
        ms, cnt = env.get_profile(1); out.append(ms / cnt * 1e3)
    return out
s = series(20); print("baseline: %s" % " ".join("%.0f" % x for x in s))
# allocate and release 100 GiB of spacers (never touched), like a long search does
lib = C.CDLL(None)
hs = []
import ctypes
hip = C.CDLL("libamdhip64.so")
class Loc(C.Structure): _fields_ = [("type", C.c_int), ("id", C.c_int)]
class Flags(C.Structure): _fields_ = [("a", C.c_ubyte), ("b", C.c_ubyte), ("c", C.c_ushort)]
class Prop(C.Structure): _fields_ = [("type", C.c_int), ("ht", C.c_int), ("loc", Loc), ("w", C.c_void_p), ("f", Flags)]
p = Prop(); p.type = 1; p.loc.type = 1; p.loc.id = 0
t0 = time.perf_counter()
for i in range(12):
    h = C.c_void_p()
    rc = hip.hipMemCreate(C.byref(h), C.c_size_t(8 << 30), C.byref(p), C.c_ulonglong(0))
    assert rc == 0, rc
    hs.append(h)
t1 = time.perf_counter()
for h in hs: hip.hipMemRelease(h)
t2 = time.perf_counter()
print("created 96 GiB in %.0f ms, released in %.0f ms" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3))
t0 = time.perf_counter()
s = series(120)
print("after release (%.0f ms of stepping): %s" % ((time.perf_counter() - t0) * 1e3, " ".join("%.0f" % x for x in s)))
PY
