"""CPU: whole-episode STRUCTURAL replays of the reference's three OLDER recordings (docs/assets/{mortar_mayhem,mystery_path,
searing_spotlights}_0.gif + searing_spotlights_0_gt.gif, SCALE 1.0; tests/golden/old_gif_replays.npz made by
tests/golden/make_old_gif_replays.py: every frame + the scene recovered from its pixels).

Those episodes come from an older revision of the reference (other defaults, other RNG order, an older pygame), so their
DYNAMICS cannot be replayed -- but the drawing of a given scene has not changed for the primitives they are made of.  Every
frame's scene (agent position / sprite, tile states and target, command glyph, fall-off cross, coins, closed / OPEN exit,
the lit discs of the spotlight layer, the dim ramp's alpha) is handed to the oracle's own drawing code
(oracle/mgo_env.h: mgo_vtbl.scene) at SCALE 1.0, which must reproduce the frame:

  mortar_mayhem_0      275 frames: arena, tile toggling (6 frames on / 18 off, the target tile stays blue), all glyphs shown,
                       agent sprite -- 0 px off; masked: the sprite box in 19 frames with on-the-fly rotated diagonal sprites
                       (that revision rotated the surface every frame, README.md:368) and <= 24 px on the "stay" glyph's
                       thick circle (older pygame: other inner edge)
  mystery_path_0       423 frames: start / goal tiles, agent sprite, the 45-degree thick-line cross in all 8 fall-off frames
                       -- 0 px off, nothing masked
  searing_spotlights_0 118 frames, observation AND ground-truth view: chessboards (blue / red), the dark layer's alpha ramp
                       (42 per step), filled discs (the lit region of every frame is exactly a union of pygame discs of
                       radius 20..64), coin, the closed exit and -- from frame 53 on -- the OPEN exit (48, 141, 70), seen
                       through the holes in the observation and over the dark layer in the ground-truth view; masked: the top
                       bar (another layout in that revision), the hand / coin rings (thick circles), the agent in the
                       observation's frame 0 (that revision's reset frame shows none)
"""
import os
import zlib

import numpy as np
import pytest

import oracle_lib

Z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "old_gif_replays.npz"))
BODY, HAND, OUTLINE = (250, 204, 153), (250, 250, 250), (50, 50, 50)
DIM = 336


def frames(key):
    pal, blob, shape = Z[key + "_pal"], Z[key + "_idx"], tuple(int(v) for v in Z[key + "_shape"])
    return pal[np.frombuffer(zlib.decompress(blob.tobytes()), np.uint8).reshape(shape[:-1])]  # [k][y][x][c]


def is_col(f, c):
    return (f == np.array(c, np.uint8)).all(-1)


def img(obs):
    return obs.transpose(1, 0, 2)


def test_mortar_mayhem_every_frame():
    fr, scenes = frames("mm"), Z["mm_scenes"]
    assert len(fr) == 275 == len(scenes)
    e = oracle_lib.OracleEnv("MortarMayhem-v0", 1.0)
    e.reset(0)
    masked = on_frames = glyph_frames = 0
    glyphs = set()
    for k, (f, v) in enumerate(zip(fr, scenes)):
        ax, ay, rot, on, tx, ty, glyph, miss = [int(x) for x in v]
        got = img(e.scene([ax, ay, rot, on, tx, ty, glyph]))
        diff = (got != f).any(-1)
        if glyph == 4:
            ring = np.zeros_like(diff)
            ring[124:212, 124:212] = True
            ring &= is_col(got, (255, 255, 255)) | is_col(f, (255, 255, 255))
            assert int((diff & ring).sum()) <= 24
            diff &= ~ring
        if miss:
            diff[max(ay - 50, 0):ay + 50, max(ax - 50, 0):ax + 50] = False
            masked += 1
        assert diff.sum() == 0, "mortar_mayhem_0 frame %d: %d px differ (scene %s)" % (k, diff.sum(), v.tolist())
        on_frames += on
        if glyph >= 0:
            glyph_frames += 1
            glyphs.add(glyph)
    e.close()
    assert masked == 19 and on_frames == 55 and glyph_frames == 30 and len(glyphs) >= 6


def test_mystery_path_every_frame():
    fr, scenes = frames("mp"), Z["mp_scenes"]
    assert len(fr) == 423 == len(scenes)
    e = oracle_lib.OracleEnv("MysteryPath-v0", 1.0)
    e.reset(0)
    crosses = 0
    for k, (f, v) in enumerate(zip(fr, scenes)):
        assert int(v[12]) == 0  # every sprite of this recording matches one of the oracle's eight exactly
        got = img(e.scene([float(x) for x in v[:12]]))
        diff = (got != f).any(-1)
        assert diff.sum() == 0, "mystery_path_0 frame %d: %d px differ (scene %s)" % (k, diff.sum(), v.tolist())
        crosses += int(v[3])
    e.close()
    assert crosses == 8


def test_searing_spotlights_every_frame_both_views():
    obs, gt, scenes = frames("ss"), frames("ssgt"), Z["ss_scenes"]
    assert len(obs) == 118 == len(gt) == len(scenes)
    e = oracle_lib.OracleEnv("SearingSpotlights-v0", 1.0)
    e.reset(0)
    open_px_obs = open_px_gt = n_discs = full_discs = red_frames = 0
    for k, row in enumerate(scenes):
        v = row[~np.isnan(row)]
        bg_red, alpha, ax, ay = [int(x) for x in v[:4]]
        n_coins = int(v[8])
        coins = [(int(v[9 + 2 * c]), int(v[10 + 2 * c])) for c in range(n_coins)]
        ns = int(v[9 + 2 * n_coins])
        discs = v[10 + 2 * n_coins:].reshape(ns, 3)
        assert alpha == min(255, 42 * k)
        got = img(e.scene(v))
        dbg = e.debug_view()
        for name, a, b in (("observation", got, obs[k]), ("ground-truth view", dbg, gt[k])):
            diff = (a != b).any(-1)
            diff[:16] = False
            ring = np.zeros((DIM, DIM), bool)
            ring[max(ay - 50, 0):ay + 50, max(ax - 50, 0):ax + 50] = True
            for (cx, cy) in coins:
                ring[max(cy - 17, 0):cy + 17, max(cx - 17, 0):cx + 17] = True
            body = is_col(a, BODY) | is_col(b, BODY)
            if (k == 0 and name == "observation") or int(Z["ss_miss"][k]):  # no agent in that revision's reset frame; diagonal
                diff[max(ay - 50, 0):ay + 50, max(ax - 50, 0):ax + 50] = False  # sprites rotated on the fly: box masked
            # outside the agent / coin boxes everything must agree; inside them everything but the thick rings
            thick = is_col(a, HAND) | is_col(b, HAND) | is_col(a, OUTLINE) | is_col(b, OUTLINE) | is_col(a, (255, 165, 0)) | is_col(b, (255, 165, 0)) | \
                is_col(a, (255, 255, 0)) | is_col(b, (255, 255, 0))
            bad = int((diff & ~ring).sum()) + int((diff & body).sum()) + (int((diff & ring & ~thick).sum()) if alpha == 255 else 0)
            assert bad == 0, "searing_spotlights_0 frame %d (%s): %d px differ" % (k, name, bad)
        open_px_obs += int((is_col(obs[k], (48, 141, 70)) & is_col(got, (48, 141, 70))).sum())
        open_px_gt += int((is_col(gt[k], (48, 141, 70)) & is_col(dbg, (48, 141, 70))).sum())
        n_discs += ns
        full_discs += int(sum(1 for (x, y, r) in discs if x - r >= 0 and x + r <= DIM and y - r >= 16 and y + r <= DIM))
        red_frames += bg_red
    e.close()
    # the OPEN exit: 65 frames x 902 px over the dark layer in the ground-truth view (the last one partly under the agent), and
    # whatever the holes show of it in the observation
    assert open_px_gt > 60 * 902 and open_px_obs > 300
    assert n_discs > 250 and full_discs > 60 and red_frames >= 1
