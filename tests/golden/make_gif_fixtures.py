#!/usr/bin/env python3
"""Turn the reference's own recordings (docs/assets/{emm,ess,emp}_0.gif: seed-0 episodes rendered by the real
PyGame code at SCALE = 1.0) into pixel fixtures for the CPU oracle.

Runs ONLY in the build container (needs /root/reference and PIL):

    python tests/golden/make_gif_fixtures.py [emm] [ess] [emp]

For each GIF:
  1. decode all frames (frame k = observation after k steps, image orientation [row=y][col=x]);
  2. recover the action stream by replaying the UNMODIFIED reference logic under the shims of ref_shims.py at
     SCALE 1.0 and, per step, keeping the action(s) whose predicted state explains the next frame;
  3. store: the actions, every frame as palette indices (+ palette), the reference's final info.
Fixtures are data only (decoded recordings + recovered inputs); no reference source is stored.
"""
import copy
import importlib
import os
import sys
import zlib

import numpy as np
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, "/root/reference")
import ref_shims  # noqa: E402

ref_shims.install()
import memory_gym  # noqa: E402,F401

ASSETS = "/root/reference/docs/assets"
BODY = (250, 204, 153)


def decode(path):
    im = Image.open(path)
    frames = []
    for k in range(im.n_frames):
        im.seek(k)
        frames.append(np.asarray(im.convert("RGB")).copy())
    return np.stack(frames)  # [k][y][x][c]


def pack_frames(frames):
    """frames [K,H,W,3] -> palette [P,3] u8 + zlib'ed index array."""
    flat = frames.reshape(-1, 3)
    key = flat[:, 0].astype(np.uint32) << 16 | flat[:, 1].astype(np.uint32) << 8 | flat[:, 2]
    pal, idx = np.unique(key, return_inverse=True)
    assert len(pal) < 256
    palette = np.stack([(pal >> 16) & 255, (pal >> 8) & 255, pal & 255], 1).astype(np.uint8)
    blob = zlib.compress(idx.astype(np.uint8).tobytes(), 9)
    return palette, np.frombuffer(blob, np.uint8), frames.shape


_YY, _XX = np.mgrid[0:336, 0:336]


def agent_matches(frame, cx, cy):
    """Is the agent body (even-diameter disc r=25 at SCALE 1, partly hidden by its two hands) centred at (cx, cy)?
    Every body-coloured pixel must lie inside the predicted disc and the visible disc area must be explained."""
    body = np.all(frame == np.array(BODY, np.uint8), axis=2)
    d2 = (_XX - cx + 0.5) ** 2 + (_YY - cy + 0.5) ** 2
    inside_loose = d2 <= 26.0 ** 2
    inside_core = d2 <= 14.0 ** 2   # hands never reach closer than 15.4 px to the centre
    if (body & ~inside_loose).any():
        return False
    if (inside_core & ~body).any():
        return False
    predicted = int((d2 <= 24.5 ** 2).sum())
    got = int(body.sum())
    return predicted - 700 <= got <= predicted + 120


def make_env(module, cls, scale_attr=True):
    mod = importlib.import_module("memory_gym." + module)
    mod.SCALE = 1.0
    return getattr(mod, cls)()


def emm():
    frames = decode(os.path.join(ASSETS, "emm_0.gif"))
    env = make_env("endless_mortar_mayhem", "EndlessMortarMayhemEnv")
    opts = {"agent_scale": 1.0, "agent_speed": 12.0}
    env.reset(seed=0, options=opts)
    # beam search: usually one hypothesis; partially off-screen agents can be ambiguous for a few frames
    beams = [(env, [])]
    info = None
    for k in range(1, len(frames)):
        new, seen = [], set()
        for e, acts in beams:
            display = bool(e._command_visualization)  # agent frozen, possibly hidden behind the glyph
            cand_actions = [(0, 0)] if display else [(a0, a1) for a0 in range(3) for a1 in range(3)]
            for a in cand_actions:
                e2 = copy.deepcopy(e)
                _, r, d, _, inf = e2.step(np.array(a))
                if d and k != len(frames) - 1:
                    continue
                cx, cy = e2.rotated_agent_rect.center
                if not display and not agent_matches(frames[k], cx, cy):
                    continue
                key = (e2.agent.rect.center, e2.agent.rotation, d)
                if key in seen:
                    continue
                seen.add(key)
                new.append((e2, acts + [a], d, inf))
        assert new, "frame %d: no action explains the agent position" % k
        beams = [(e, a) for (e, a, d, i) in new[:16]]
        last = new
    finals = [(e, a, i) for (e, a, d, i) in last if d]
    assert finals, "reference logic did not terminate on the GIF's last frame"
    env, actions, info = finals[0]
    done = True
    print("emm: surviving hypotheses at the end:", len(last))
    assert done, "reference logic did not terminate on the GIF's last frame"
    print("emm: %d actions; final info %s" % (len(actions), {k: (v.tolist() if hasattr(v, 'tolist') else v) for k, v in info.items()}))
    palette, blob, shape = pack_frames(frames)
    np.savez(os.path.join(HERE, "gif_emm_0.npz"), env_id="Endless-MortarMayhem-v0", seed=0, actions=np.array(actions, np.int8),
             palette=palette, frames_zlib=blob, frames_shape=np.array(shape),
             final_reward=info["reward"], final_length=info["length"], commands_completed=info["commands_completed"],
             max_command_sequence=info["max_command_sequence"])


def ess():
    frames = decode(os.path.join(ASSETS, "ess_0.gif"))
    # frame k+1's top bar shows action k (grey 0 / purple 1 / orange 2), sampled at (x=200,y=2) and (x=300,y=2)
    colors = {(120, 120, 120): 0, (116, 1, 113): 1, (255, 94, 14): 2}
    actions = []
    for k in range(1, len(frames) - 1):
        actions.append((colors[tuple(frames[k + 1][2, 200])], colors[tuple(frames[k + 1][2, 300])]))
    env = make_env("endless_searing_spotlights", "EndlessSearingSpotlightsEnv")
    opts = {"agent_scale": 1.0, "agent_speed": 12.0, "coin_scale": 1.5, "spot_min_radius": 30.0, "spot_max_radius": 55.0,
            "agent_health": 20}
    env.reset(seed=0, options=opts)
    done = False
    for k, a in enumerate(actions):
        _, r, done, _, info = env.step(np.array(a))
        assert not done, "terminated early at step %d" % (k + 1)
    # the last action is not shown by the recording (the top bar lags one frame): keep the ones that end the episode
    # on the last frame, as the recording does
    last = []
    for a0 in range(3):
        for a1 in range(3):
            e2 = copy.deepcopy(env)
            _, r, d, _, info = e2.step(np.array([a0, a1]))
            if d:
                last.append((a0, a1))
    assert last, "no final action terminates the episode on the recording's last frame"
    print("ess: terminating final actions:", last)
    actions.append(last[0])
    _, r, done, _, info = env.step(np.array(last[0]))
    print("ess: %d actions; final info %s" % (len(actions), {k: (v.tolist() if hasattr(v, 'tolist') else v) for k, v in info.items()}))
    palette, blob, shape = pack_frames(frames)
    np.savez(os.path.join(HERE, "gif_ess_0.npz"), env_id="Endless-SearingSpotlights-v0", seed=0,
             actions=np.array(actions, np.int8), palette=palette, frames_zlib=blob, frames_shape=np.array(shape),
             agent_health=20, final_reward=info["reward"], final_length=info["length"], coins_collected=info["coins_collected"])


def emp_features(frame):
    """Observable features of an Endless-MysteryPath frame at SCALE 1.0: agent body rows, red cross present,
    mask of white past-path tile pixels."""
    body = np.all(frame == np.array(BODY, np.uint8), axis=2)
    red = np.all(frame == np.array((255, 0, 0), np.uint8), axis=2)
    white = np.all(frame == np.array((255, 255, 255), np.uint8), axis=2)
    ys = np.nonzero(body.any(1))[0]
    return (int(ys.min()) if len(ys) else -1, int(ys.max()) if len(ys) else -1, bool(red.any()), white)


def emp_predict(env):
    """What the reference state predicts for the same features (geometry from endless_mystery_path.py:111-160,232-242)."""
    td = env.tile_dim
    cy = env.agent.rect.center[1]
    top, bot = cy - 25, cy + 24
    cross = env.fall_off_surface.get_alpha() == 255
    return top, bot, cross


def emp():
    frames = decode(os.path.join(ASSETS, "emp_0.gif"))
    env = make_env("endless_mystery_path", "EndlessMysteryPathEnv")
    opts = {"agent_scale": 1.0, "agent_speed": 12.0}
    env.reset(seed=0, options=opts)

    def past_tiles(e):
        """set of (draw_x, y) of the past-path tiles the reference would draw (endless_mystery_path.py:111-132)"""
        out = []
        x = e.normalized_agent_position[0] - 1
        if x < 0:
            return out
        depth = int(e.reset_params["camera_offset_scale"])
        past_x = max(0, x - depth)
        node = e.current_node.previous_node
        while x >= past_x and x >= 0:
            if node is None:
                break
            x, y = node.x, node.y
            out.append((int(x * e.tile_dim - e.camera_x), int(y * e.tile_dim)))
            if x == past_x:
                break
            node = node.previous_node
        return out

    def explains(e, frame):
        top, bot, cross, white = emp_features(frame)
        ptop, pbot, pcross = emp_predict(e)
        # the agent may be partly hidden by the cross (drawn on top, centred on the agent): compare body extent loosely
        if top >= 0 and not (abs(top - max(ptop, 0)) <= 0 or cross):
            return False
        if bool(cross) != bool(pcross):
            return False
        td = e.tile_dim
        pred = np.zeros(frame.shape[:2], bool)
        for (dx, dy) in past_tiles(e):
            x0, x1 = max(dx + 1, 0), min(dx + td - 1, frame.shape[1])
            y0, y1 = max(dy + 1, 0), min(dy + td - 1, frame.shape[0])
            if x1 > x0 and y1 > y0:
                pred[y0:y1, x0:x1] = True
        # compare only tile interiors not covered by agent/cross
        covered = ~np.all(frame == 0, axis=2) & ~white
        agree = (white == pred) | covered
        return bool(agree.all())

    beams = [(env, [])]
    for k in range(1, len(frames)):
        new = []
        seen = set()
        for e, acts in beams:
            for a in range(4):
                e2 = copy.deepcopy(e)
                _, r, d, _, info = e2.step(a)
                if d and k != len(frames) - 1:
                    continue
                if explains(e2, frames[k]):
                    key = (e2.agent.rect.center, e2.camera_x, e2.is_off_path, e2.stamina, e2.max_x_reached,
                           tuple(e2.fall_off_locations), e2.tiles_visited)
                    if key in seen:
                        continue
                    seen.add(key)
                    new.append((e2, acts + [a], d, info))
        assert new, "frame %d: no hypothesis survives" % k
        if len(new) > 8:
            new = new[:8]
        beams = [(e, a) for (e, a, d, i) in new]
        last = new
    finals = [(e, a, i) for (e, a, d, i) in last if d]
    assert finals, "no hypothesis terminates on the last frame"
    e, actions, info = finals[0]
    print("emp: %d actions, %d surviving hypotheses; final info %s" % (len(actions), len(finals),
          {k: (v.tolist() if hasattr(v, 'tolist') else v) for k, v in info.items()}))
    palette, blob, shape = pack_frames(frames)
    np.savez(os.path.join(HERE, "gif_emp_0.npz"), env_id="Endless-MysteryPath-v0", seed=0, actions=np.array(actions, np.int8),
             palette=palette, frames_zlib=blob, frames_shape=np.array(shape), final_reward=info["reward"],
             final_length=info["length"], num_fails=info["num_fails"], max_x=info["max_x"], tiles_visited=info["tiles_visited"])


if __name__ == "__main__":
    which = sys.argv[1:] or ["emm", "ess", "emp"]
    for w in which:
        {"emm": emm, "ess": ess, "emp": emp}[w]()
        fn = os.path.join(HERE, "gif_%s_0.npz" % w)
        print("  ->", fn, os.path.getsize(fn) // 1024, "KiB")
