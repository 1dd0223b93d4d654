#!/usr/bin/env python3
"""Fixtures from the reference's three OLDER recordings (docs/assets/{searing_spotlights,mortar_mayhem,mystery_path}_0.gif:
SCALE 1.0, an older revision of the code -- other defaults and RNG order -- rendered by an older pygame; SURVEY.md section 4
and App. E: filled circles, rects, rounded rects, glyph lines/rotations are still identical, the inner edge of THICK
circles is not).  They cannot be replayed, but single frames pin what no other reference artefact shows: the finite
environments' layouts and the `Exit` stamp.  Runs ONLY in the build container (needs /root/reference and PIL):

    python tests/golden/make_old_gif_fixtures.py

Stored in tests/golden/old_gifs.npz (data only: decoded frames / crops as palette indices):
  ss_frame0, mm_frame0, mp_frame0   frame 0 of each recording, [y][x] palette indices + palette
  mm_glyphs                         the distinct command-glyph crops of mortar_mayhem_0.gif (display frames in which the
                                    agent does not touch the 88x88 centre box)
"""
import os
import zlib

import numpy as np
from PIL import Image

ASSETS = "/root/reference/docs/assets"
HERE = os.path.dirname(os.path.abspath(__file__))
BODY = (250, 204, 153)


def decode(path):
    im = Image.open(path)
    frames = []
    for k in range(im.n_frames):
        im.seek(k)
        frames.append(np.asarray(im.convert("RGB")).copy())
    return np.stack(frames)  # [k][y][x][c]


def pack(frames):
    flat = frames.reshape(-1, 3)
    key = flat[:, 0].astype(np.uint32) << 16 | flat[:, 1].astype(np.uint32) << 8 | flat[:, 2]
    pal, idx = np.unique(key, return_inverse=True)
    assert len(pal) < 256
    palette = np.stack([(pal >> 16) & 255, (pal >> 8) & 255, pal & 255], 1).astype(np.uint8)
    return palette, np.frombuffer(zlib.compress(idx.astype(np.uint8).tobytes(), 9), np.uint8), np.array(frames.shape)


def main():
    out = {}
    for key, name in (("ss", "searing_spotlights_0"), ("mm", "mortar_mayhem_0"), ("mp", "mystery_path_0")):
        fr = decode(os.path.join(ASSETS, name + ".gif"))
        pal, blob, shape = pack(fr[:1])
        out[key + "_frame0_pal"], out[key + "_frame0_idx"], out[key + "_frame0_shape"] = pal, blob, shape
        print(name, fr.shape)
        if key == "mm":
            crops = []
            for f in fr:
                c = f[124:212, 124:212]
                if (c == np.array(BODY)).all(-1).any() or (c == 255).all(-1).sum() < 50:
                    continue  # the agent reaches into the box, or no glyph is shown
                if not any(np.array_equal(c, q) for q in crops):
                    crops.append(c.copy())
            print("distinct glyph crops:", len(crops))
            pal, blob, shape = pack(np.stack(crops))
            out["mm_glyphs_pal"], out["mm_glyphs_idx"], out["mm_glyphs_shape"] = pal, blob, shape
    np.savez(os.path.join(HERE, "old_gifs.npz"), **out)
    print("wrote", os.path.join(HERE, "old_gifs.npz"), os.path.getsize(os.path.join(HERE, "old_gifs.npz")), "bytes")


if __name__ == "__main__":
    main()
