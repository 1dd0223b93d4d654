"""CPU test of the product's HOST-side stamp builder (csrc/mg_stamps.hpp) over the *_scale reset options: for a sweep of
agent_scale / coin_scale / exit_scale values the sprites, the coin and the exit (closed and open) the library would upload,
blitted by a numpy model of the composers, must equal what the oracle draws for the same scene (its `scene` test hook:
chessboard, coin, exit, agent, no dark layer).  The sweep crosses every width regime of pygame.draw.circle (filled, the 1-px
circle of width == 1, thick rings) and of the exit's outline."""
import os
import struct
import subprocess

import numpy as np
import pytest

import oracle_lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PALETTE = {1: (250, 204, 153), 2: (250, 250, 250), 3: (50, 50, 50), 4: (255, 255, 255), 5: (255, 0, 0), 8: (255, 255, 0), 9: (255, 165, 0),
           15: (0, 0, 0), 16: (48, 141, 70), 17: (55, 55, 55)}


@pytest.fixture(scope="module")
def dumper(tmp_path_factory):
    d = tmp_path_factory.mktemp("stamps")
    exe = str(d / "dump_stamps")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-o", exe, os.path.join(ROOT, "tools", "dump_stamps.cpp")])

    def dump(agent_scale, coin_scale, exit_scale):
        out = str(d / "s.bin")
        subprocess.check_call([exe, repr(agent_scale), "2", out, repr(coin_scale), repr(exit_scale)])
        b = open(out, "rb").read()
        D, radius, ng, n = struct.unpack_from("4i", b, 0)
        off = 16
        sprites = np.frombuffer(b, np.uint8, 8 * D * D, off).reshape(8, D, D)  # [k][y][x]
        off += 8 * D * D
        for _ in range(ng):
            w, h = struct.unpack_from("2i", b, off)
            off += 8 + w * h
        off += (1 + n * n) * 84 * 84 * 3
        extra = []
        for _ in range(3):
            w, h = struct.unpack_from("2i", b, off)
            off += 8
            extra.append(np.frombuffer(b, np.uint8, w * h, off).reshape(h, w))
            off += w * h
        return sprites, extra[0], extra[1], extra[2]

    return dump


def blit(frame, stamp, x0, y0):
    """stamp [y][x] of palette ids (0 = colour key) with its top-left at (x0, y0) into frame [x][y][c]"""
    for py in range(stamp.shape[0]):
        for px in range(stamp.shape[1]):
            idx = int(stamp[py, px])
            X, Y = x0 + px, y0 + py
            if idx and 0 <= X < 84 and 0 <= Y < 84:
                frame[X, Y] = PALETTE[idx]


SCALES = [(0.1, 0.15, 0.2), (0.125, 0.1875, 0.25), (0.2, 0.3, 0.3), (0.25, 0.375, 0.5), (0.28, 0.45, 0.55), (0.3, 0.5, 0.6), (0.34, 0.55, 0.7),
          (0.4, 0.6, 0.75), (0.45, 0.7, 0.9), (0.5, 0.75, 1.0), (0.55, 0.9, 1.1), (0.6, 1.0, 1.25), (0.67, 1.2, 1.5), (0.8, 1.5, 1.7), (1.0, 2.0, 2.0)]


@pytest.mark.parametrize("agent_scale,coin_scale,exit_scale", SCALES)
def test_scaled_stamps_equal_the_oracles_drawing(dumper, agent_scale, coin_scale, exit_scale):
    sprites, coin, exit_closed, exit_open = dumper(agent_scale, coin_scale, exit_scale)
    e = oracle_lib.OracleEnv("SearingSpotlights-v0")
    e.reset(1, options=dict(agent_scale=agent_scale, coin_scale=coin_scale, exit_scale=exit_scale))
    # v = {bg_red, alpha, ax, ay, sprite, exit_x, exit_y, exit_open, n_coins, (x, y) * n_coins, n_spots, ...}
    empty = e.scene([0, 0, -300, -300, 0, -300, -300, 0, 0, 0])
    D = sprites.shape[1]
    r = coin.shape[0] // 2 if coin.shape[0] > 1 else 0
    half = exit_closed.shape[0] >> 1
    for sprite in range(8):
        for open_ in (0, 1):
            ax, ay, cx, cy, ex, ey = 20 + sprite, 60 - sprite, 62, 24 + sprite, 58 - sprite, 62
            got = e.scene([sprite & 1, 0, ax, ay, sprite, ex, ey, open_, 1, cx, cy, 0])
            base = e.scene([sprite & 1, 0, -300, -300, 0, -300, -300, 0, 0, 0]) if sprite & 1 else empty
            want = base.copy()
            blit(want, coin, cx - r, cy - r)
            blit(want, exit_open if open_ else exit_closed, ex - half, ey - half)
            blit(want, sprites[sprite], ax - D // 2, ay - D // 2)
            bad = np.argwhere((got[:, 4:] != want[:, 4:]).any(-1))
            assert len(bad) == 0, "scales %s sprite %d open %d: %d px differ below the top bar, first (x, y - 4) = %s" % (
                (agent_scale, coin_scale, exit_scale), sprite, open_, len(bad), bad[0])
    e.close()
