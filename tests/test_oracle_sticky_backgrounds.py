"""CPU oracle vs tests/golden/sticky_backgrounds.npz, captured from the unmodified reference by
tests/golden/make_sticky_fixture.py: one environment object per id driven through several episodes with different
option dictionaries; after every call the fixture holds what hide_chessboard / black_background have left of the two
background surfaces, which of them is shown and each live spotlight's has_border
(/root/reference/memory_gym/searing_spotlights.py:349-351, 234-235, 420-421; pygame_assets.py:62)."""
import importlib.util
import json
import os

import numpy as np
import pytest

import oracle_lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FIX = os.path.join(ROOT, "tests", "golden", "sticky_backgrounds.npz")


def full_options(env_id, opts):
    """the complete dictionary the reference's process_reset_params() would build (module loaded without the HIP library)"""
    spec = importlib.util.spec_from_file_location("rp", os.path.join(ROOT, "endless-memory-gym_amd", "memory_gym_amd", "reset_params.py"))
    rp = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(rp)
    return rp.process_reset_params(env_id, opts)


@pytest.mark.parametrize("tag,env_id", [("ss_", "SearingSpotlights-v0"), ("ess_", "Endless-SearingSpotlights-v0")])
def test_board_states_and_borders_follow_the_reference(tag, env_id):
    z = np.load(FIX)
    g = {k[len(tag):]: z[k] for k in z.files if k.startswith(tag)}
    options = json.loads(str(g["options"]))
    env = oracle_lib.OracleEnv(env_id, scale=0.25)
    n = len(g["kind"])
    assert n > 300 and g["done"].sum() >= 3 and set(np.unique(g["blue"])) == {0, 1, 2} and (g["borders"] == 1).any()
    for k in range(n):
        if g["kind"][k] == 0:
            seed = int(g["seed"][k])
            env.reset(None if seed < 0 else seed, options=full_options(env_id, options[int(g["phase"][k])]), want_obs=False)
        else:
            _, r, d = env.step([int(g["a0"][k]), int(g["a1"][k])], want_obs=False)
            assert r == g["reward"][k] and d == bool(g["done"][k]), "row %d" % k
        assert env.get("bg_blue_mode") == g["blue"][k], "row %d: blue board" % k
        assert env.get("bg_red_mode") == g["red"][k], "row %d: red board" % k
        assert env.get("bg_red") == g["bg_red"][k], "row %d: board shown" % k
        want = g["borders"][k]
        want = want[want >= 0]
        got = env.get_list("borders")
        assert len(got) == len(want) and np.array_equal(got.astype(np.int8), want), "row %d: borders" % k
    env.close()
