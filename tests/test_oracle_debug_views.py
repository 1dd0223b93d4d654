"""CPU: render("debug_rgb_array") of the oracle against the reference's debug-view recordings docs/assets/{emm,ess,emp}_0_gt.gif
(tests/golden/gif_*_0_gt.npz: every third frame + the first 40, made by tests/golden/make_gt_gif_fixtures.py), replayed
with the action streams recovered from the observation recordings (tests/golden/gif_*_0.npz), SCALE 1.0."""
import ctypes as C
import os
import zlib

import numpy as np
import pytest

import oracle_lib

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def debug_view(env):
    L = oracle_lib.lib()
    L.mgo_render_debug.argtypes = [C.c_void_p, C.c_void_p]
    out = np.zeros((336, 336, 3), np.uint8)
    assert L.mgo_render_debug(env.h, out.ctypes.data) == 0
    return out


def gt(name):
    z = np.load(os.path.join(GOLDEN, "gif_%s_0_gt.npz" % name))
    shape = tuple(int(v) for v in z["frames_shape"])
    idx = np.frombuffer(zlib.decompress(z["frames_zlib"].tobytes()), np.uint8).reshape(shape[:3])
    return {int(k): z["palette"][idx[j]] for j, k in enumerate(z["frame_numbers"])}, int(z["total_frames"])


@pytest.mark.parametrize("name,options", [("emm", None), ("ess", dict(agent_health=20)), ("emp", None)])
def test_debug_view_equals_the_recording(name, options):
    z = np.load(os.path.join(GOLDEN, "gif_%s_0.npz" % name))
    frames, total = gt(name)
    env = oracle_lib.OracleEnv(str(z["env_id"]), scale=1.0)
    env.reset(int(z["seed"]), options=options)
    checked, bad = 0, []
    for k in range(total):
        if k:
            env.step(z["actions"][k - 1])
        view = debug_view(env)  # every frame, like the recording loop: each render pops the mortar family's clone list
        if k not in frames:
            continue
        if name == "ess" and k == total - 1:
            continue  # the recording's last action is unknowable (tests/test_oracle_gif.py)
        d = view != frames[k]
        if name == "emp":
            # the stamina bar (drawn in the debug view only) of the recording regains a point on the respawn step after a
            # fall; the reference's current step() excludes the start tile there (endless_mystery_path.py:347) -- the
            # recording predates that line, the logic fixtures (tests/golden/logic_*.npz) follow the current code
            d[:, 320:] = False
        checked += 1
        if d.any():
            bad.append((k, int(d.any(2).sum())))
    assert checked > 190 and not bad, "%d of %d debug frames differ from the recording, first %s" % (len(bad), checked, bad[:8])


def test_debug_view_is_the_debug_surface_stretched_to_336():
    """SCALE 0.25: every pixel of the 84x84 debug surface becomes a 4x4 block (pygame.transform.scale = transform.c stretch())."""
    env = oracle_lib.OracleEnv("MysteryPath-v0", 0.25)
    env.reset(3)
    for a in ([2, 0], [2, 0], [0, 2]):
        env.step(a)
    d = debug_view(env)
    small = d[::4, ::4]
    assert np.array_equal(np.repeat(np.repeat(small, 4, 0), 4, 1), d)
    assert (small == np.array((255, 255, 255))).all(-1).sum() >= 3 * 144  # path tiles are visible in the debug view
    assert (small == np.array((255, 0, 0))).all(-1).sum() >= 144           # and so are the walls
