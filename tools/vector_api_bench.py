#!/usr/bin/env python3
"""tools/vector_api_bench.py [ENV_ID N STEPS] -- what the gymnasium-0.29 vector convention costs: env-steps/s through
memory_gym_amd.vector.GymnasiumVectorEnv (terminal observations kept in infos["final_observation"], sub-environments reset in the same call)
against VecMemoryGym.step with auto-reset (what bench.py measures), same instances, same uniform random actions, torch tensors throughout."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "endless-memory-gym_amd"))
import memory_gym_amd  # noqa: E402
from memory_gym_amd.vector import GymnasiumVectorEnv  # noqa: E402

env_id = sys.argv[1] if len(sys.argv) > 1 else "MortarMayhem-Grid-v0"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 300


FOLLOW = os.environ.get("SOAK_POLICY") == "follower"  # ids whose ground truth names the way: the agent follows it (eps 0.02), episodes get long


def run(label, make, step):
    env = make()
    adim = env.env.action_dim if hasattr(env, "env") else env.action_dim
    n_act = 4 if adim == 1 else 3
    env.reset(seed=0)
    g = torch.Generator(device="cuda").manual_seed(1)
    acts = [torch.randint(0, n_act, (n,) if adim == 1 else (n, adim), device="cuda", generator=g, dtype=torch.int32) for _ in range(16)]
    inner = env.env if hasattr(env, "env") else env
    follow = FOLLOW and inner.gt_dim == 3
    rnd = [torch.rand(n, device="cuda", generator=g) < 0.02 for _ in range(16)]

    def act(t, gt):
        return torch.where(rnd[t % 16], acts[t % 16], gt.argmax(1).to(torch.int32) + 1) if follow else acts[t % 16]
    gt = inner.gt.clone()
    for t in range(600 if follow else 100):
        out = step(env, act(t, gt))
        gt = out[4]["ground_truth"] if follow else gt
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for t in range(steps):
        out = step(env, act(t, gt))
        gt = out[4]["ground_truth"] if follow else gt
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("%-52s %7.1f M env-steps/s  %7.1f us per step" % (label, n * steps / dt / 1e6, dt / steps * 1e6), flush=True)
    env.close()


run("%s x %d, VecMemoryGym.step (auto-reset)" % (env_id, n), lambda: memory_gym_amd.make(env_id, num_envs=n, device=0), lambda e, a: e.step(a))
run("... GymnasiumVectorEnv.step (final_observation)", lambda: GymnasiumVectorEnv(env_id, n, device=0), lambda e, a: e.step(a))
