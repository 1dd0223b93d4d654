#!/usr/bin/env python3
"""use_exit = False on an environment object that has had an exit before, captured from the unmodified reference (build
container only).

SearingSpotlights' reset() spawns no exit when `use_exit` is False (searing_spotlights.py:413-416) -- no position is sampled, no
number drawn -- but its frame still blits `self.exit` (:431-435, :534-536): the Exit object of an EARLIER episode, at its place
and in the state (open / closed) it was last drawn in; the episode then ends with the last coin (:499-511).  (On an object that
never had an exit the same line raises AttributeError: that case stays refused.)  One environment object is driven through
phases with use_exit on and off; after every call the fixture holds reward, done, terminal info, the PCG64 words and the
exit's position and state.

    python tests/golden/make_stale_exit_fixture.py      # -> tests/golden/stale_exit.npz
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as G  # noqa: E402  (installs the shims, imports the reference)

ENV_ID = "SearingSpotlights-v0"
EASY = dict(agent_health=200, initial_spawns=1, num_spawns=2)
PHASES = [
    (dict(EASY, num_coins=[1], max_steps=90), 1.0, 140),                      # exits get opened and reached: the stale exit is OPEN
    (dict(EASY, use_exit=False, num_coins=[2], max_steps=70), 1.0, 200),      # ... episodes end with the last coin
    (dict(EASY, num_coins=[3], max_steps=12), 0.0, 40),                       # cut short: the exit stays CLOSED
    (dict(EASY, use_exit=False, num_coins=[1, 2], max_steps=50, exit_visible=True, coins_visible=True), 0.9, 160),
    (dict(EASY, num_coins=[1], max_steps=90), 1.0, 80),
]


def main():
    spec = G.ENVS[ENV_ID]
    env = G.make(ENV_ID)
    prng = np.random.Generator(np.random.PCG64(20260929))
    rows = []

    def record(kind, phase, seed, action, reward, done, info):
        rows.append(dict(kind=kind, phase=phase, seed=seed, a0=int(action[0]), a1=int(action[1]), reward=float(reward), done=int(done),
                         exit_x=int(env.exit.location[0]), exit_y=int(env.exit.location[1]), exit_open=int(env.exit.open),
                         n_coins_left=len(env.coins), success=float(info.get("success", -1)) if done else -1.0,
                         info_reward=float(info.get("reward", np.nan)) if done else np.nan, rng=G.rng_words(env)))

    options = []
    for ph, (opts, skill, steps) in enumerate(PHASES):
        options.append(dict(opts))
        seed = 500 + 13 * ph
        _, info = env.reset(seed=seed, options=opts)
        record(0, ph, seed, (0, 0), 0.0, 0, info)
        for _ in range(steps):
            a = np.asarray(spec["pol"](env, prng, skill))
            _, r, done, _, info = env.step(a)
            record(1, ph, -1, a, r, done, info)
            if done:
                _, info = env.reset(options=opts)
                record(0, ph, -1, (0, 0), 0.0, 0, info)
    out = {k: np.array([r[k] for r in rows]) for k in rows[0] if k != "rng"}
    out["rng"] = np.stack([r["rng"] for r in rows])
    out["options"] = np.array(json.dumps(options))
    ph = out["phase"]
    off = np.isin(ph, [1, 3])
    print("rows", len(rows), "episodes", int(out["done"].sum()), "episodes without an exit", int(out["done"][off].sum()),
          "rows with a stale OPEN exit", int((out["exit_open"][off] == 1).sum()), "with a stale CLOSED exit", int((out["exit_open"][off] == 0).sum()))
    fn = os.path.join(HERE, "stale_exit.npz")
    np.savez_compressed(fn, **out)
    print("->", fn, os.path.getsize(fn) // 1024, "KiB")


if __name__ == "__main__":
    main()
