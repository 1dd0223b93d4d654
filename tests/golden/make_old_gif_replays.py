#!/usr/bin/env python3
"""Whole-episode STRUCTURAL replays of the reference's three OLDER recordings (docs/assets/{mortar_mayhem,mystery_path,
searing_spotlights}_0.gif and searing_spotlights_0_gt.gif: SCALE 1.0, an older revision of the reference -- other defaults and
RNG order, an older pygame).  Their DYNAMICS cannot be replayed, but the DRAWING of a given
scene has not changed for the primitives these frames are made of: arena tiles and their toggling, command glyphs, start /
goal tiles, the fall-off cross (45-degree thick lines), the chessboards, the alpha ramp of the dark layer, filled discs
(spotlight holes), the coin, the closed AND the open Exit.

For every frame this script recovers the SCENE from the pixels (agent position and sprite, tile states and target tile,
glyph, cross, coins, exit, the lit discs) and stores it next to the frame; tests/test_oracle_old_gif_replay.py puts the
oracle into each scene (oracle/mgo_env.h: mgo_vtbl.scene) and requires its own drawing code to reproduce the frame.

Runs ONLY in the build container (needs /root/reference, PIL, scipy and the oracle):

    python tests/golden/make_old_gif_replays.py        ->  tests/golden/old_gif_replays.npz

The fixture is DATA: decoded frames (palette indices, zlib) and the recovered scene vectors.
"""
import os
import sys
import zlib

import numpy as np
from PIL import Image
from scipy import ndimage, signal

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import oracle_lib  # noqa: E402

ASSETS = "/root/reference/docs/assets"
BODY, HAND, OUTLINE = (250, 204, 153), (250, 250, 250), (50, 50, 50)
DIM = 336


def decode(name):
    im = Image.open(os.path.join(ASSETS, name + ".gif"))
    frames = []
    for k in range(im.n_frames):
        im.seek(k)
        frames.append(np.asarray(im.convert("RGB")).copy())
    return np.stack(frames)  # [k][y][x][c]


def pack(frames):
    flat = frames.reshape(-1, 3)
    key = flat[:, 0].astype(np.uint32) << 16 | flat[:, 1].astype(np.uint32) << 8 | flat[:, 2]
    pal, idx = np.unique(key, return_inverse=True)
    assert len(pal) < 256, len(pal)
    palette = np.stack([(pal >> 16) & 255, (pal >> 8) & 255, pal & 255], 1).astype(np.uint8)
    return palette, np.frombuffer(zlib.compress(idx.astype(np.uint8).tobytes(), 9), np.uint8), np.array(frames.shape)


def is_col(f, c):
    return (f == np.array(c, np.uint8)).all(-1)


def img(obs):  # array3d [x][y][c] -> image [y][x][c]
    return obs.transpose(1, 0, 2)


# ---- the agent: which of the eight sprites, where -----------------------------------------------------------------
class SpriteMatcher:
    """Body-colour masks of the oracle's eight sprites (SCALE 1.0), matched against a frame's body-coloured pixels."""

    def __init__(self):
        e = oracle_lib.OracleEnv("MysteryPath-v0", 1.0)
        e.reset(0)
        self.masks = []
        for rot in range(8):
            f = img(e.scene([168, 168, rot, 0, 0, 0, 0, 0, 6, 6, 0, 0]))
            self.masks.append(is_col(f, BODY)[168 - 50:168 + 50, 168 - 50:168 + 50])
        e.close()

    def find(self, body, occluded=None):
        """body: bool [y][x]; occluded: pixels where the body may be hidden.  Returns (ax, ay, rot, mismatch)."""
        ys, xs = np.nonzero(body)
        if len(ys) == 0:
            return None
        cy, cx = int(round(ys.mean())), int(round(xs.mean()))
        pad = np.zeros((DIM + 200, DIM + 200), bool)
        pad[100:100 + DIM, 100:100 + DIM] = body
        occ = np.zeros_like(pad)
        if occluded is not None:
            occ[100:100 + DIM, 100:100 + DIM] = occluded
        outside = np.ones_like(pad)
        outside[100:100 + DIM, 100:100 + DIM] = False
        best = None
        for rot in range(8):
            m = self.masks[rot]
            for ay in range(cy - 14, cy + 15):
                for ax in range(cx - 14, cx + 15):
                    win = (slice(ay + 100 - 50, ay + 100 + 50), slice(ax + 100 - 50, ax + 100 + 50))
                    care = ~(occ[win] | outside[win])
                    miss = int(((pad[win] != m) & care).sum())
                    if best is None or miss < best[3]:
                        best = (ax, ay, rot, miss)
        # pixels of the body outside the 100x100 window would be a mismatch as well
        return best


# ---- Mortar Mayhem -------------------------------------------------------------------------------------------------
def replay_mm(frames, sm):
    e = oracle_lib.OracleEnv("MortarMayhem-v0", 1.0)
    e.reset(0)
    # glyph crops of the oracle (centre box 88 x 88 at 124), the agent parked in a corner tile
    glyph_box = (slice(124, 212), slice(124, 212))
    crops = {g: img(e.scene([60, 60, 0, 0, 0, 0, g]))[glyph_box].copy() for g in range(10)}
    arena_x0, tile = (DIM - 5 * 56) // 2, 56
    scenes, worst, rotated = [], 0, 0
    for k, f in enumerate(frames):
        red = is_col(f, (81, 18, 26)) | is_col(f, (112, 24, 36))
        on = int(red.any())
        tx = ty = 0
        if on:  # the one tile that stays blue
            blue_tiles = []
            for i in range(5):
                for j in range(5):
                    t = f[arena_x0 + j * tile: arena_x0 + (j + 1) * tile, arena_x0 + i * tile: arena_x0 + (i + 1) * tile]
                    if not (is_col(t, (81, 18, 26)) | is_col(t, (112, 24, 36))).any():
                        blue_tiles.append((i, j))
            assert len(blue_tiles) == 1, (k, blue_tiles)
            tx, ty = blue_tiles[0]
        white = is_col(f, (255, 255, 255))
        ax, ay, rot, miss = sm.find(is_col(f, BODY), occluded=white)
        glyph = -1
        if white.any():
            cands = []
            for g in range(9):
                gm = is_col(crops[g], (255, 255, 255))
                fm = white[glyph_box]
                cands.append((int((gm != fm).sum()), g))
            glyph = min(cands)[1]
        v = [ax, ay, rot, on, tx, ty, glyph, miss]
        got = img(e.scene(v))
        diff = (got != f).any(-1)
        hands = is_col(got, HAND) | is_col(got, OUTLINE) | is_col(f, HAND) | is_col(f, OUTLINE)
        if glyph == 4:  # "stay" = a THICK circle + bar: the older pygame's inner edge differs in a few pixels
            ring = np.zeros_like(diff)
            ring[glyph_box] = True
            ring &= is_col(got, (255, 255, 255)) | is_col(f, (255, 255, 255))
            assert int((diff & ring).sum()) <= 24, (k, int((diff & ring).sum()))
            diff &= ~ring
        if miss:  # that revision rotated the sprite's surface every frame (README.md:368): its diagonal sprites are others
            diff[max(ay - 50, 0):ay + 50, max(ax - 50, 0):ax + 50] = False
            rotated += 1
        n_bad = int((diff & ~hands).sum())
        worst = max(worst, int((diff & hands).sum()))
        assert n_bad == 0, "mortar_mayhem_0 frame %d: %d px differ outside the hand rings (scene %s, body mismatch %d)" % (k, n_bad, v, miss)
        scenes.append(v)
    e.close()
    print("mortar_mayhem_0: %d frames reproduced (%d with the sprite box masked); hand-ring pixels differing per frame <= %d" % (len(frames), rotated, worst))
    return np.array(scenes, np.int16)


# ---- Mystery Path --------------------------------------------------------------------------------------------------
def replay_mp(frames, sm):
    e = oracle_lib.OracleEnv("MysteryPath-v0", 1.0)
    e.reset(0)
    g = np.nonzero(is_col(frames[0], (0, 255, 0)))
    b = np.nonzero(is_col(frames[1], (0, 0, 255)) | is_col(frames[0], (0, 0, 255)))
    ex, ey = int(g[1].min()) // 48, int(g[0].min()) // 48
    # the blue start tile is partly under the agent in frame 0: take its cell from any pixel of it
    sx, sy = int(b[1].min()) // 48, int(b[0].min()) // 48
    scenes, worst, rotated = [], 0, 0
    for k, f in enumerate(frames):
        cross = is_col(f, (255, 0, 0))
        found = sm.find(is_col(f, BODY), occluded=cross)
        cross_on = int(cross.any())
        ax, ay, rot, miss = found
        cx = cy = 0
        if cross_on:  # the cross is drawn at the agent's rect centre; its own bounding box gives the centre as well
            ys, xs = np.nonzero(cross)
            cx, cy = (int(xs.min()) + int(xs.max()) + 1) // 2, (int(ys.min()) + int(ys.max()) + 1) // 2
        v = [ax, ay, rot, cross_on, cx, cy, sx, sy, ex, ey, 1, 1, miss]
        got = img(e.scene(v))
        diff = (got != f).any(-1)
        hands = is_col(got, HAND) | is_col(got, OUTLINE) | is_col(f, HAND) | is_col(f, OUTLINE)
        if miss:
            box = np.zeros_like(diff)
            box[max(ay - 50, 0):ay + 50, max(ax - 50, 0):ax + 50] = True
            diff &= ~(box & ~(cross | is_col(got, (255, 0, 0))))  # the cross stays judged
            rotated += 1
        n_bad = int((diff & ~hands).sum())
        worst = max(worst, int((diff & hands).sum()))
        assert n_bad == 0, "mystery_path_0 frame %d: %d px differ outside the hand rings (scene %s, body mismatch %d)" % (k, n_bad, v, miss)
        scenes.append(v)
    e.close()
    print("mystery_path_0: %d frames reproduced (%d with the sprite box masked); hand-ring pixels differing per frame <= %d" % (len(frames), rotated, worst))
    return np.array(scenes, np.int16)


# ---- Searing Spotlights -------------------------------------------------------------------------------------------
_DISCS = {}


def disc(r):
    """pygame's filled circle of radius r (the oracle's rasteriser) as a bool [2r][2r] stamp; pixel (r, r) is the centre"""
    if r not in _DISCS:
        L = oracle_lib.lib()
        import ctypes as C
        L.mgo_test_circle.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
        d = 2 * r + 4
        out = np.zeros((d, d), np.uint8)
        assert L.mgo_test_circle(d, r + 2, r + 2, r, 0, out.ctypes.data) == 0
        _DISCS[r] = out.astype(bool)
    return _DISCS[r]


PAD = 70


def paint(shape, x, y, r):
    m = np.zeros(shape, bool)
    d = disc(r)
    y0, x0 = y - r - 2 + PAD, x - r - 2 + PAD
    ya, xa = max(y0, 0), max(x0, 0)
    yb, xb = min(y0 + d.shape[0], shape[0]), min(x0 + d.shape[1], shape[1])
    if ya < yb and xa < xb:
        m[ya:yb, xa:xb] = d[ya - y0:yb - y0, xa - x0:xb - x0]
    return m


def fit_discs(lit, care):
    """Explain the lit region (bool [y][x]; `care` marks the pixels that can be judged) as a union of pygame discs, every
    frame on its own: repeatedly the disc (radius 64 .. 20, centre anywhere, also off screen) that lies entirely inside the
    lit / unjudged / off-screen region and explains most of the largest still unexplained blob, the LARGEST such radius.
    A disc that is fully on screen has exactly one explanation; a cap at the border has several, any of which will do."""
    shape = (DIM + 2 * PAD, DIM + 2 * PAD)
    L = np.zeros(shape, bool)
    L[PAD:PAD + DIM, PAD:PAD + DIM] = lit
    K = np.zeros(shape, bool)  # judged pixels
    K[PAD:PAD + DIM, PAD:PAD + DIM] = care
    free = L | ~K              # where a disc may lie
    discs = []
    covered = np.zeros(shape, bool)
    guard = 0
    while (L & K & ~covered).any():
        guard += 1
        assert guard < 16, "disc fit does not converge"
        rest = L & K & ~covered
        lab, n = ndimage.label(rest)
        sizes = ndimage.sum(rest, lab, range(1, n + 1))
        blob = lab == (1 + int(np.argmax(sizes)))
        ys, xs = np.nonzero(blob)
        y0, y1 = max(int(ys.min()) - 68, 0), min(int(ys.max()) + 69, shape[0])
        x0, x1 = max(int(xs.min()) - 68, 0), min(int(xs.max()) + 69, shape[1])
        # the crop's own border counts as "may lie here": a disc reaching past it is checked again on the full canvas below
        fr = np.pad(free[y0:y1, x0:x1], 70, constant_values=True).astype(np.float32)
        bl = np.pad(blob[y0:y1, x0:x1], 70, constant_values=False).astype(np.float32)
        best = None
        for r in range(64, 19, -1):
            d = disc(r).astype(np.float32)
            area = float(d.sum())
            inside = signal.fftconvolve(fr, d[::-1, ::-1], mode="same")
            ok = inside > area - 0.5
            if not ok.any():
                continue
            hits = signal.fftconvolve(bl, d[::-1, ::-1], mode="same")
            hits[~ok] = -1
            order = np.argsort(hits, axis=None)[::-1][:6]
            for flat in order:
                iy, ix = np.unravel_index(int(flat), hits.shape)
                if hits[iy, ix] <= 0:
                    break
                # 'same' centres the kernel at its middle element; try the neighbouring alignments on the exact canvas
                for dy in range(-2, 3):
                    for dx in range(-2, 3):
                        x, y = ix - 70 + x0 - PAD + dx, iy - 70 + y0 - PAD + dy
                        m = paint(shape, x, y, r)
                        if (m & ~free).any():
                            continue
                        g = int((m & blob).sum())
                        if best is None or g > best[0]:
                            best = (g, r, x, y)
        assert best is not None and best[0] > 0, "no disc explains the remaining lit pixels"
        _, r, x, y = best
        discs.append((x, y, r))
        covered |= paint(shape, x, y, r)
    assert not (covered & ~free).any()
    return discs


def replay_ss(obs_frames, gt_frames, sm):
    e = oracle_lib.OracleEnv("SearingSpotlights-v0", 1.0)
    e.reset(0)
    scenes, lens, prev, misses = [], [], [], []
    worst = 0
    # where the exit's fill lies relative to the exit position (rect centre), measured on the oracle's own stamp
    probe = img(e.scene([0, 0, 30, 300, 0, 168, 168, 0, 0, 0]))
    py, px = np.nonzero(is_col(probe, (55, 55, 55)))
    fill_dx, fill_dy = 168 - int(px.min()), 168 - int(py.min())
    for k, (f, g) in enumerate(zip(obs_frames, gt_frames)):
        alpha = min(255, 42 * k)
        # the ground-truth view shows the agent, the coins and the exit over the dark layer
        ax, ay, rot, miss = sm.find(is_col(g, BODY))
        closed, opened = is_col(g, (55, 55, 55)), is_col(g, (48, 141, 70))
        fill = opened if opened.any() else closed
        is_open = int(opened.any())
        ys, xs = np.nonzero(fill)
        exit_x, exit_y = int(xs.min()) + fill_dx, int(ys.min()) + fill_dy
        coins = []
        yel = is_col(g, (255, 255, 0))
        if yel.any():
            lab, n = ndimage.label(yel | is_col(g, (255, 165, 0)))
            for c in range(1, n + 1):
                cy, cx = np.nonzero(lab == c)
                coins.append(((int(cx.min()) + int(cx.max()) + 1) // 2, (int(cy.min()) + int(cy.max()) + 1) // 2))
        board_red = int(is_col(f[16:], (255, 0, 0)).any())
        # lit pixels of the observation (below the top bar): anything that is not the dark layer's black; the exit's own
        # black border cannot be judged
        care = np.ones((DIM, DIM), bool)
        care[:16] = False
        lit = np.zeros((DIM, DIM), bool)
        if alpha == 255:
            lit = ~is_col(f, (0, 0, 0))
            lit[:16] = False
            care[max(exit_y - 21, 0):exit_y + 21, max(exit_x - 21, 0):exit_x + 21] &= lit[max(exit_y - 21, 0):exit_y + 21, max(exit_x - 21, 0):exit_x + 21]
            prev = fit_discs(lit, care)
        elif alpha > 0:
            bright = is_col(f, (255, 255, 255)) | is_col(f, (0, 0, 255))
            bright[:16] = False
            assert not bright.any(), "a spotlight is on screen during the dim ramp (frame %d)" % k
        v = [board_red, alpha, ax, ay, rot, exit_x, exit_y, is_open, len(coins)] + [c for xy in coins for c in xy] + [len(prev)] + [c for d in prev for c in d]
        try:
            got = img(e.scene(v))
        except AssertionError:
            print("frame", k, "scene refused:", v)
            raise
        dbg = e.debug_view()
        for name, a, b in (("observation", got, f), ("ground-truth view", dbg, g)):
            diff = (a != b).any(-1)
            diff[:16] = False  # the top bar of that revision is another one (no last-reward bar, other action colours)
            ring = np.zeros((DIM, DIM), bool)  # thick circles of the older pygame: hand rings, coin ring
            ring[max(ay - 50, 0):ay + 50, max(ax - 50, 0):ax + 50] = True
            for (cx, cy) in coins:
                ring[max(cy - 17, 0):cy + 17, max(cx - 17, 0):cx + 17] = True
            body = is_col(a, BODY) | is_col(b, BODY)
            if (k == 0 and name == "observation") or miss:  # that revision's reset frame does not show the agent; its diagonal
                diff[max(ay - 50, 0):ay + 50, max(ax - 50, 0):ax + 50] = False  # sprites were rotated on the fly (README.md:368)
            n_bad = int((diff & ~ring).sum()) + int((diff & body).sum())
            worst = max(worst, int((diff & ring & ~body).sum()))
            assert n_bad == 0, "searing_spotlights_0 frame %d (%s): %d px differ (scene %s, body mismatch %d)" % (k, name, n_bad, v, miss)
        scenes.append(v)
        lens.append(len(v))
        misses.append(miss)
    e.close()
    print("searing_spotlights_0: %d frames reproduced (observation and ground-truth view); ring pixels differing per frame <= %d; "
          "discs per frame up to %d" % (len(scenes), worst, max((len(s) - 10) // 3 for s in scenes)))
    out = np.full((len(scenes), max(lens)), np.nan)
    for i, s in enumerate(scenes):
        out[i, :len(s)] = s
    return out, np.array(misses, np.int16)


def main():
    sm = SpriteMatcher()
    out = {}
    mm = decode("mortar_mayhem_0")
    out["mm_scenes"] = replay_mm(mm, sm)
    out["mm_pal"], out["mm_idx"], out["mm_shape"] = pack(mm)
    mp = decode("mystery_path_0")
    out["mp_scenes"] = replay_mp(mp, sm)
    out["mp_pal"], out["mp_idx"], out["mp_shape"] = pack(mp)
    ss, gt = decode("searing_spotlights_0"), decode("searing_spotlights_0_gt")
    out["ss_scenes"], out["ss_miss"] = replay_ss(ss, gt, sm)
    out["ss_pal"], out["ss_idx"], out["ss_shape"] = pack(ss)
    out["ssgt_pal"], out["ssgt_idx"], out["ssgt_shape"] = pack(gt)
    fn = os.path.join(HERE, "old_gif_replays.npz")
    np.savez(fn, **out)
    print("wrote", fn, os.path.getsize(fn) // 1024, "KiB")


if __name__ == "__main__":
    main()
