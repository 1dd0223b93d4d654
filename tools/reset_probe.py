import sys, time
sys.path.insert(0, "endless-memory-gym_amd")
import torch, memory_gym_amd
for env_id, n in [("MysteryPath-v0", 32768), ("Endless-MysteryPath-v0", 32768), ("MysteryPath-Grid-v0", 32768), ("Endless-SearingSpotlights-v0", 16384), ("MortarMayhem-Grid-v0", 65536)]:
    e = memory_gym_amd.make(env_id, num_envs=n, device=0)
    e.reset(seed=0); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(5): e.reset(seed=k * n)
    torch.cuda.synchronize()
    print(env_id, n, "full reset (logic + raster): %.0f us" % ((time.perf_counter() - t0) / 5 * 1e6))
    e.close()
