#!/usr/bin/env python3
"""Fixtures for render("debug_rgb_array"): the reference's recordings docs/assets/{emm,ess,emp}_0_gt.gif show the DEBUG view
(`_build_debug_surface`) of the same three seed-0 episodes whose observations are in {emm,ess,emp}_0.gif (SCALE 1.0).
Stored per recording: every third frame plus the first 40 (palette indices, zlib) -- replayed with the action streams of
tests/golden/gif_*_0.npz.  Runs ONLY in the build container (needs /root/reference and PIL):

    python tests/golden/make_gt_gif_fixtures.py
"""
import os
import zlib

import numpy as np
from PIL import Image

ASSETS = "/root/reference/docs/assets"
HERE = os.path.dirname(os.path.abspath(__file__))


def decode(path):
    im = Image.open(path)
    frames = []
    for k in range(im.n_frames):
        im.seek(k)
        frames.append(np.asarray(im.convert("RGB")).copy())
    return np.stack(frames)


for name in ("emm", "ess", "emp"):
    fr = decode(os.path.join(ASSETS, name + "_0_gt.gif"))
    keep = sorted(set(range(0, len(fr), 3)) | set(range(min(40, len(fr)))) | {len(fr) - 1})
    sel = fr[keep]
    flat = sel.reshape(-1, 3)
    key = flat[:, 0].astype(np.uint32) << 16 | flat[:, 1].astype(np.uint32) << 8 | flat[:, 2]
    pal, idx = np.unique(key, return_inverse=True)
    assert len(pal) < 256
    palette = np.stack([(pal >> 16) & 255, (pal >> 8) & 255, pal & 255], 1).astype(np.uint8)
    out = os.path.join(HERE, "gif_%s_0_gt.npz" % name)
    np.savez(out, palette=palette, frames_zlib=np.frombuffer(zlib.compress(idx.astype(np.uint8).tobytes(), 9), np.uint8),
             frames_shape=np.array(sel.shape), frame_numbers=np.array(keep), total_frames=len(fr))
    print(name, fr.shape, "kept", len(keep), os.path.getsize(out), "bytes")
