"""CPU: pin what the three v1.0 recordings cannot show -- the FINITE environments' layouts and the `Exit` stamp -- to
frame 0 of the reference's three OLDER recordings (tests/golden/old_gifs.npz, decoded by tests/golden/
make_old_gif_fixtures.py from docs/assets/{searing_spotlights,mortar_mayhem,mystery_path}_0.gif, SCALE 1.0).  Those episodes
come from an older revision (other defaults, other RNG order, an older pygame whose THICK circles have another inner edge:
SURVEY.md section 4 / App. E), so the comparison is masked: everything except the agent sprite, the coin's ring and the
command glyph's identity must agree pixel for pixel with the oracle at SCALE 1.0."""
import os
import zlib

import numpy as np
import pytest

import oracle_lib

Z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "old_gifs.npz"))


def unpack(key):
    pal, blob, shape = Z[key + "_pal"], Z[key + "_idx"], tuple(Z[key + "_shape"])
    idx = np.frombuffer(zlib.decompress(blob.tobytes()), np.uint8).reshape(shape[:-1])
    return pal[idx]  # [k][y][x][c]


def oracle_frame(env_id, seed, options=None):
    e = oracle_lib.OracleEnv(env_id, 1.0)
    f = e.reset(seed, options=options).transpose(1, 0, 2)  # array3d [x][y] -> image [y][x]
    return e, f


def test_searing_spotlights_layout_and_exit():
    """chessboard, health / reward bars, the Exit (closed: fill (55,55,55), 8-px black border, top corners rounded with
    radius 40 -- pygame_assets.py:169-204, restated from pygame's draw_round_rect) and its sampled position, the coin's
    position and outer edge: identical to the recording.  Masked: the agent (the old frame 0 shows none) and the coin's ring."""
    gif = unpack("ss_frame0")[0]
    e, f = oracle_frame("SearingSpotlights-v0", 0)
    closed = (gif == 55).all(-1)
    ys, xs = np.nonzero(closed)
    assert closed.sum() == 902 and (xs.min(), xs.max(), ys.min(), ys.max()) == (146, 177, 44, 75)  # the recording's exit
    box = (slice(ys.min() - 8, ys.max() + 13), slice(xs.min() - 8, xs.max() + 9))  # fill + border
    assert np.array_equal(f[box], gif[box]), "Exit stamp differs from the recording"
    diff = (f != gif).any(-1)
    ax, ay = int(e.get("ax")), int(e.get("ay"))
    diff[max(ay - 60, 0):ay + 60, max(ax - 60, 0):ax + 60] = False  # agent sprite (absent from the old frame 0)
    coin = (gif == np.array((255, 255, 0))).all(-1)
    cy, cx = [int(round(v.mean())) for v in np.nonzero(coin)]
    ring = np.hypot(*np.mgrid[0:336, 0:336][::-1] - np.array([cx, cy])[:, None, None])
    assert np.array_equal(f[ring <= 8], gif[ring <= 8])  # coin centre: same place, same yellow
    diff[ring <= 17] = False  # the coin's thick ring: older pygame, other inner edge (App. E)
    assert diff.sum() == 0, "%d px differ outside the masked agent / coin ring" % diff.sum()


def test_mortar_mayhem_layout_and_glyphs():
    """arena offset, tile size / colours / borders and the agent's sampled position equal the recording; every command glyph
    the recording shows (7 distinct ones) equals one of the oracle's nine glyphs pixel for pixel."""
    gif = unpack("mm_frame0")[0]
    e, f = oracle_frame("MortarMayhem-v0", 0)
    diff = (f != gif).any(-1)
    ax, ay = int(e.get("ax")), int(e.get("ay"))
    body = (gif == np.array((250, 204, 153))).all(-1)
    assert body[ay, ax], "the agent's sampled position differs from the recording"
    diff[ay - 40:ay + 40, ax - 40:ax + 40] = False  # sprite: body edge and hand rings (thick circles) differ in the old pygame
    diff[124:212, 124:212] = False                   # which command is shown first (older RNG order)
    assert diff.sum() == 0, "%d px differ outside the masked agent / glyph boxes" % diff.sum()
    # glyph rasters: collect the oracle's centre box for every command over a few episodes
    seen = {}
    for seed in range(40):
        e, f = oracle_frame("MortarMayhem-v0", seed)
        for t in range(70):
            g = int(e.get("glyph"))
            c = f[124:212, 124:212]
            ax, ay = e.get("ax"), e.get("ay")
            if g >= 0 and g not in seen and max(abs(ax - 168), abs(ay - 168)) > 44 + 42:  # the sprite (hands included) stays out of the box
                seen[g] = c.copy()
            o, _, done = e.step([0, 0])
            f = o.transpose(1, 0, 2)
            if done or len(seen) == 9:
                break
        if len(seen) == 9:
            break
    assert len(seen) >= 9
    crops = unpack("mm_glyphs")
    assert len(crops) == 7
    exact, ring = set(), 0
    for k, c in enumerate(crops):
        miss, g = min(((c != s).any(-1).sum(), g) for g, s in seen.items())
        if miss == 0:
            exact.add(g)
        else:  # "stay" = a THICK circle + bar: the older pygame's inner edge differs in a few pixels, nothing else may
            assert miss <= 24, "glyph crop %d of the recording is none of the oracle's glyphs (closest: %d, %d px)" % (k, g, miss)
            ring += 1
    assert len(exact) == 6 and ring == 1  # six arrows (axis-parallel and diagonal) pixel-exact, plus the stay glyph


def test_mystery_path_tiles():
    """start / goal tile rects (48 px at SCALE 1.0; the recording was made when they were shown by default): position on the
    7x7 grid, size and colours -- found with the options that show them and a seed whose path has the recording's ends."""
    gif = unpack("mp_frame0")[0]
    green = (gif == np.array((0, 255, 0))).all(-1)
    blue = (gif == np.array((0, 0, 255))).all(-1)
    gy, gx = np.nonzero(green)
    by, bx = np.nonzero(blue)
    assert (gx.min(), gx.max(), gy.min(), gy.max()) == (0, 47, 240, 287) and (bx.min(), bx.max(), by.min(), by.max()) == (288, 335, 144, 191)
    opts = dict(show_origin=True, show_goal=True)
    for seed in range(2000):
        e, f = oracle_frame("MysteryPath-v0", seed, opts)
        p = e.get_list("path")
        if p is not None and len(p) >= 4 and tuple(p[:2]) == (0, 5) and tuple(p[-2:]) == (6, 3):
            break
    else:
        pytest.fail("no seed below 2000 has the recording's path ends")
    g2 = (f == np.array((0, 255, 0))).all(-1)
    b2 = (f == np.array((0, 0, 255))).all(-1)
    ax, ay = int(e.get("ax")), int(e.get("ay"))
    keep = np.ones((336, 336), bool)
    keep[max(ay - 40, 0):ay + 40, max(ax - 40, 0):ax + 40] = False  # the agent stands on the start tile
    assert np.array_equal(b2 & keep, blue & keep) and np.array_equal(g2 & keep, green & keep)
    diff = (f != gif).any(-1) & keep
    diff[0:40, 0:40] = False  # the recording's agent (older start logic: top-left corner)
    assert diff.sum() == 0, "%d px differ outside the two agent boxes" % diff.sum()
