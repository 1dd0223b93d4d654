import sys, os, time
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT","/root/repo"), "endless-memory-gym_amd"))
import numpy as np, torch, memory_gym_amd
env = memory_gym_amd.make("MortarMayhem-Grid-v0")
env.reset(seed=1)
g = np.random.Generator(np.random.PCG64(12345))
acts = g.integers(0, 4, 20000)
# full adapter
t0=time.perf_counter(); n=0
for a in acts:
    o,r,d,_,i = env.step(int(a)); n+=1
    if d: env.reset()
dt=time.perf_counter()-t0
print("adapter step(): %.1f k steps/s (%.2f us)" % (n/dt/1e3, dt/n*1e6))
# native call only
f, h, rs = env._step_fn, env._h, env._raw_stream
env.reset(seed=1)
t0=time.perf_counter()
for a in acts:
    f(h, 0, 0, rs())
dt=time.perf_counter()-t0
print("mg_single_step only (noop action, no reset): %.1f k/s (%.2f us)" % (len(acts)/dt/1e3, dt/len(acts)*1e6))
s = rs()
t0=time.perf_counter()
for a in acts:
    f(h, 0, 0, s)
dt=time.perf_counter()-t0
print("... with the stream looked up once: %.1f k/s (%.2f us)" % (len(acts)/dt/1e3, dt/len(acts)*1e6))
t0=time.perf_counter()
for a in acts:
    env._observation()
dt=time.perf_counter()-t0
print("_observation() copy: %.2f us" % (dt/len(acts)*1e6))
