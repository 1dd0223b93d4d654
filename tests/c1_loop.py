#!/usr/bin/env python3
"""BASELINE config C1 (the reference's own bench.py loop, /root/reference/bench.py:12-30): MortarMayhem-Grid-v0, ONE
instance, reset(seed=1), 1000 random-action episodes with reset() inside the timed region; prints mean +- std of the
per-episode steps/s and the mean success.  Actions come from the fixed stream Generator(PCG64(12345)) so that every
backend sees the same episode sequence.

    python tests/c1_loop.py --backend oracle      # CPU restatement (any host)
    python tests/c1_loop.py --backend hip         # single-instance adapter over libmemgym_hip.so (MI355X; latency-bound)
    python tests/c1_loop.py --backend reference   # the PyGame reference, only if memory-gym is installed on the host
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))  # tests/ -> repo root
sys.path.insert(0, os.path.join(ROOT, "endless-memory-gym_amd"))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def make(backend, env_id):
    if backend == "oracle":
        import oracle_lib

        class E:
            def __init__(self):
                self.e = oracle_lib.OracleEnv(env_id)
                self.success = 0.0

            def reset(self, seed=None):
                return self.e.reset(seed), {}

            def step(self, a):
                o, r, d = self.e.step([a, 0])
                info = {"success": self.e.get("info_success")} if d else {}
                return o, r, d, False, info
        return E()
    if backend == "hip":
        import memory_gym_amd
        return memory_gym_amd.make(env_id)
    import gymnasium as gym
    import memory_gym  # noqa: F401
    return gym.make(env_id)


def run(backend, env_id="MortarMayhem-Grid-v0", episodes=1000):
    """The C1 recipe; returns {"steps_per_s_mean", "steps_per_s_std", "episodes", "steps", "mean_success", "checksum"} --
    `checksum` = the total number of steps, which every backend must agree on (same seed, same action stream)."""
    env = make(backend, env_id)
    g = np.random.Generator(np.random.PCG64(12345))
    fps, succ, total = [], [], 0
    seed = 1
    for ep in range(episodes):
        t0 = time.perf_counter()
        env.reset(seed=seed)
        seed = None  # later resets continue the env's RNG stream
        done, steps = False, 0
        while not done:
            _, _, done, _, info = env.step(int(g.integers(0, 4)))
            steps += 1
        fps.append(steps / (time.perf_counter() - t0))
        succ.append(float(info.get("success", 0)))
        total += steps
    if hasattr(env, "close"):
        env.close()
    return {"backend": backend, "steps_per_s_mean": float(np.mean(fps)), "steps_per_s_std": float(np.std(fps)), "episodes": episodes,
            "steps": total, "mean_success": float(np.mean(succ))}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--backend", default="oracle", choices=["oracle", "hip", "reference"])
    ap.add_argument("--env", default="MortarMayhem-Grid-v0")
    ap.add_argument("--episodes", type=int, default=1000)
    ap.add_argument("--json", action="store_true")
    args = ap.parse_args()
    r = run(args.backend, args.env, args.episodes)
    if args.json:
        import json
        print(json.dumps(r))
        return
    print("backend %s  %s  episodes %d  steps %d  mean steps/s %.1f  std %.1f  mean success %.3f" % (
        args.backend, args.env, args.episodes, r["steps"], r["steps_per_s_mean"], r["steps_per_s_std"], r["mean_success"]))


if __name__ == "__main__":
    main()
