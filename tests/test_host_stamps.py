"""CPU test of the product's HOST-side stamp/template builder (csrc/mg_stamps.hpp): a numpy model of the raster
kernel's composition (template -> sprite -> glyph) must reproduce the oracle's frames bit-exactly."""
import os
import struct
import subprocess

import numpy as np
import pytest

import oracle_lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PALETTE = np.array([[0, 0, 0], [250, 204, 153], [250, 250, 250], [50, 50, 50], [255, 255, 255], [255, 0, 0]], np.uint8)


def dump(agent_scale, N, tmp_path):
    exe = str(tmp_path / "dump_stamps")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-o", exe, os.path.join(ROOT, "tools", "dump_stamps.cpp")])
    out = str(tmp_path / "stamps.bin")
    subprocess.check_call([exe, str(agent_scale), str(N), out])
    b = open(out, "rb").read()
    D, radius, ng, n = struct.unpack_from("4i", b, 0)
    off = 16
    sprites = np.frombuffer(b, np.uint8, 8 * D * D, off).reshape(8, D, D)  # [k][y][x]
    off += 8 * D * D
    glyphs = []
    for _ in range(ng):
        w, h = struct.unpack_from("2i", b, off)
        off += 8
        glyphs.append(np.frombuffer(b, np.uint8, w * h, off).reshape(h, w))
        off += w * h
    templ = np.frombuffer(b, np.uint8, (1 + n * n) * 84 * 84 * 3, off).reshape(1 + n * n, 84, 84, 3)
    return sprites, glyphs, templ, radius


def compose(templ, sprites, glyphs, tmpl, sprite, cx, cy, glyph):
    f = templ[tmpl].copy()  # [x][y][c]
    D = sprites.shape[1]
    for py in range(D):
        for px in range(D):
            idx = sprites[sprite][py, px]
            X, Y = cx - D // 2 + px, cy - D // 2 + py
            if idx and 0 <= X < 84 and 0 <= Y < 84:
                f[X, Y] = PALETTE[idx]
    if 0 <= glyph < 9:
        g = glyphs[glyph]
        for py in range(g.shape[0]):
            for px in range(g.shape[1]):
                if g[py, px]:
                    f[31 + px, 31 + py] = (255, 255, 255)
    return f


@pytest.mark.parametrize("env_id,N,opts", [("MortarMayhem-Grid-v0", 5, None), ("MortarMayhem-v0", 3, dict(arena_size=3)),
                                           ("Endless-MortarMayhem-v0", 6, None)])
def test_composed_frames_equal_oracle(env_id, N, opts, tmp_path):
    sprites, glyphs, templ, _ = dump(0.25, N, tmp_path)
    e = oracle_lib.OracleEnv(env_id)
    prng = np.random.Generator(np.random.PCG64(1))
    for seed in range(6):
        obs = e.reset(seed, options=opts)
        for t in range(120):
            if t:
                a = [int(prng.integers(0, 4)), 0] if e.discrete else prng.integers(0, 3, 2)
                obs, _, done = e.step(a)
            tiles_on = int(e.get("tiles_on")) and 1
            tmpl = 1 + int(e.get("tx")) * N + int(e.get("ty")) if tiles_on else 0
            if t == 0:
                sprite, cx, cy = 0, int(e.get("ax")), int(e.get("ay"))
            else:
                sprite, cx, cy = int(e.get("disp_sprite")), int(e.get("disp_x")), int(e.get("disp_y"))
            exp = compose(templ, sprites, glyphs, tmpl, sprite, cx, cy, int(e.get("glyph")))
            assert np.array_equal(exp, obs), "%s seed %d step %d" % (env_id, seed, t)
            if t and done:
                break
