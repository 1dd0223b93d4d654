#!/usr/bin/env python3
"""What hide_chessboard / black_background leave behind, captured from the unmodified reference (build container only).

The reference's SearingSpotlights environments repaint the two background surfaces they create once in __init__
(searing_spotlights.py:349-351, 234-235, 420-421; endless_searing_spotlights.py:313-315, 223-224, 376-377) and give
spotlights spawned under black_background a border (pygame_assets.py:62).  Under the pygame shim a Surface has no pixels,
but it can remember the colour of its last fill(): this script runs ONE environment object per id through several
episodes with different (complete) option dictionaries and records, after every call, the last fill of both boards, which
board is shown and every live spotlight's has_border.

    python tests/golden/make_sticky_fixture.py      # -> tests/golden/sticky_backgrounds.npz

The fixture is data: seeds, options, actions in; rewards, dones, board states and border flags out.
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as G  # noqa: E402  (installs the shims, imports the reference)
import ref_shims  # noqa: E402


def _fill(self, *a, **k):  # instrumentation of OUR shim: remember the colour
    c = a[0] if a else k.get("color")
    self._last_fill = tuple(c) if isinstance(c, (tuple, list)) else (int(c),)


ref_shims.Surface.fill = _fill


def board_mode(surf):
    c = getattr(surf, "_last_fill", None)
    if c is None:
        return 0
    if all(v == 255 for v in c) and len(c) == 3:
        return 1
    if all(v == 0 for v in c):
        return 2
    raise AssertionError("unexpected fill %r" % (c,))


FAST = dict(spot_min_speed=0.01, spot_max_speed=0.03, initial_spawns=4, agent_health=6, max_steps=48)
PHASES = [
    (dict(), 30),
    (dict(FAST, black_background=True, light_dim_off_duration=0, light_threshold=150), 90),
    (dict(FAST, hide_chessboard=True, light_dim_off_duration=2), 60),
    (dict(FAST, black_background=True, visual_feedback=False), 60),
    (dict(), 60),
    (dict(FAST, hide_chessboard=True, black_background=True), 60),
]
MAX_SPOTS = 24


def run(env_id):
    env = G.make(env_id)
    for s in (env.blue_background_surface, env.red_background_surface):
        s._last_fill = None  # whatever __init__ did to them: these are the chessboards
    prng = np.random.Generator(np.random.PCG64(4242))
    rows = []

    def record(kind, phase, seed, action, reward, done):
        b = [int(s.has_border) for s in env.spotlights]
        rows.append(dict(kind=kind, phase=phase, seed=seed, a0=action[0], a1=action[1], reward=float(reward), done=int(done),
                         blue=board_mode(env.blue_background_surface), red=board_mode(env.red_background_surface),
                         bg_red=int(env.bg is env.red_background_surface), borders=b + [-1] * (MAX_SPOTS - len(b))))

    options = []
    for ph, (opts, steps) in enumerate(PHASES):
        o = dict(opts)
        if env_id.startswith("Endless") and "agent_health" in o:
            o["spawn_interval"] = 8
        options.append(o)
        seed = 31 + 7 * ph
        env.reset(seed=seed, options=o)
        record(0, ph, seed, (0, 0), 0.0, 0)
        for _ in range(steps):
            a = prng.integers(0, 3, 2)
            _, r, done, _, _ = env.step(np.asarray(a))
            record(1, ph, -1, (int(a[0]), int(a[1])), r, done)
            if done:
                env.reset(options=o)
                record(0, ph, -1, (0, 0), 0.0, 0)
    out = {k: np.array([r[k] for r in rows]) for k in rows[0] if k != "borders"}
    out["borders"] = np.array([r["borders"] for r in rows], np.int8)
    out["options"] = np.array(json.dumps(options))
    return out


def main():
    res = {}
    for env_id in ("SearingSpotlights-v0", "Endless-SearingSpotlights-v0"):
        o = run(env_id)
        tag = "ess_" if env_id.startswith("Endless") else "ss_"
        for k, v in o.items():
            res[tag + k] = v
        print(env_id, "rows", len(o["kind"]), "episodes", int(o["done"].sum()), "blue modes", np.bincount(o["blue"], minlength=3),
              "red modes", np.bincount(o["red"], minlength=3), "rows with borders", int((o["borders"] == 1).any(1).sum()))
    fn = os.path.join(HERE, "sticky_backgrounds.npz")
    np.savez_compressed(fn, **res)
    print("->", fn, os.path.getsize(fn) // 1024, "KiB")


if __name__ == "__main__":
    main()
