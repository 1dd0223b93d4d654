"""CPU: pin the stamps at SCALE 0.25 -- the scale every BASELINE.json configuration runs at -- to SURVEY.md App. F (the
tables the survey derived with the restatement that reproduces the reference's SCALE-1.0 recordings pixel for pixel):
agent sprite body and the hard-coded hand positions of all eight rotations, the "right" and "stay" glyphs, tile geometry
and colours of MortarMayhem-Grid.  The product's host-side stamp builder (csrc/mg_stamps.hpp through tools/dump_stamps.cpp)
is checked here; tests/test_host_stamps.py ties the same stamps to the oracle's frames."""
import numpy as np

from test_host_stamps import dump

BODY = {8: (12, 15), 9: (11, 16), 10: (10, 17), 11: (9, 18), 12: (8, 19), 13: (8, 19), 14: (8, 19), 15: (8, 19),
        16: (9, 18), 17: (10, 17), 18: (11, 16), 19: (12, 15)}  # App. F: r=6 disc at (14,14), rows 8-19 (first, last column)
HANDS = {0: ((10, 10), (18, 10)), 1: ((8, 14), (13, 8)), 2: ((10, 18), (10, 10)), 3: ((14, 19), (8, 14)),
         4: ((18, 18), (10, 18)), 5: ((19, 14), (14, 19)), 6: ((18, 10), (18, 18)), 7: ((14, 8), (19, 14))}  # App. F table
RIGHT = """..........WW..........
...........WW.........
............WW........
.............WW.......
..............WW......
...............WW.....
................WW....
.................WW...
..................WW..
...................WW.
WWWWWWWWWWWWWWWWWWWWWW
WWWWWWWWWWWWWWWWWWWWW.
..................WW..
.................WW...
................WW....
...............WW.....
..............WW......
.............WW.......
............WW........
...........WW.........
..........WW..........
......................"""
STAY = """.......WWWWWW.........
.....WWWWWWWWWW.......
....WWWW....WWWW......
...WWW........WWW.....
..WW............WW....
.WWW............WWW...
.WW..............WW...
WWW..............WWW..
WW................WW..
WWWWWWWWWWWWWWWWWWWW..
WWWWWWWWWWWWWWWWWWWW..
WW................WW..
WWW..............WWW..
.WW..............WW...
.WWW............WWW...
..WW............WW....
...WWW........WWW.....
....WWWW....WWWW......
.....WWWWWWWWWW.......
.......WWWWWW.........
......................
......................"""


def hand_disc(cx, cy):
    """r=2 even-diameter disc: the 4x4 block around the centre without its corners (App. F, rows 8-11 of sprite 0)"""
    return {(x, y) for x in range(cx - 2, cx + 2) for y in range(cy - 2, cy + 2)} - {(cx - 2, cy - 2), (cx + 1, cy - 2), (cx - 2, cy + 1), (cx + 1, cy + 1)}


def art(s):
    return np.array([[c == "W" for c in row] for row in s.splitlines()])


def test_agent_sprites_all_rotations(tmp_path):
    sprites, glyphs, templ, radius = dump(0.25, 5, tmp_path)
    assert sprites.shape == (8, 28, 28) and radius == 6
    for k in range(8):
        want = np.zeros((28, 28), np.uint8)
        for y, (x0, x1) in BODY.items():
            want[y, x0:x1 + 1] = 1          # palette 1 = (250,204,153)
        for cx, cy in HANDS[k]:             # hands are drawn after (over) the body
            for x, y in hand_disc(cx, cy):
                want[y, x] = 3              # palette 3 = (50,50,50)
        assert np.array_equal(sprites[k], want), "sprite %d (rotation %d degrees)" % (k, 45 * k)


def test_glyphs_right_and_stay(tmp_path):
    sprites, glyphs, templ, _ = dump(0.25, 5, tmp_path)
    right = art(RIGHT)
    assert np.array_equal(glyphs[0] != 0, right)
    # left / up / down are rot90 multiples of "right" (counter-clockwise: up = 90, left = 180, down = 270)
    rots = [np.rot90(right, k) for k in range(4)]
    others = [g != 0 for g in glyphs[1:4]]
    assert all(any(np.array_equal(o, r) for r in rots[1:]) for o in others) and len({o.tobytes() for o in others}) == 3
    assert any(g.shape == (22, 22) and np.array_equal(g != 0, art(STAY)) for g in glyphs), "no glyph equals App. F's 'stay'"


def test_grid_templates(tmp_path):
    sprites, glyphs, templ, _ = dump(0.25, 5, tmp_path)
    assert templ.shape == (26, 84, 84, 3)
    fill, border, red_fill, red_border = (21, 43, 77), (29, 60, 107), (81, 18, 26), (112, 24, 36)
    base = templ[0]
    want = np.zeros((84, 84, 3), np.uint8)
    for i in range(5):
        for j in range(5):
            x0, y0 = 7 + 14 * i, 7 + 14 * j
            want[x0:x0 + 14, y0:y0 + 14] = border
            want[x0 + 1:x0 + 13, y0 + 1:y0 + 13] = fill
    assert np.array_equal(base, want)
    for tx in range(5):
        for ty in range(5):  # template 1 + tx*5 + ty: every tile red except the target (tx, ty)
            w = want.copy()
            for i in range(5):
                for j in range(5):
                    if (i, j) != (tx, ty):
                        x0, y0 = 7 + 14 * i, 7 + 14 * j
                        w[x0:x0 + 14, y0:y0 + 14] = red_border
                        w[x0 + 1:x0 + 13, y0 + 1:y0 + 13] = red_fill
            assert np.array_equal(templ[1 + tx * 5 + ty], w), (tx, ty)
