"""CPU: the order in which mg_obs_alloc_for deals the pieces of an observation buffer to the memory zones (include/memgym.h: mg_obs_plan,
the same code without allocating).  A raster launch writes at FRONTS that are one window = 14,336 observations apart (its persistent
workgroups stride over the frames); what is fast is a split of the concurrently written fronts over the zones (profiles/r02_zones.md).
Walk the launch over the plan and look at where its fronts are: for every output format and launch size, at (nearly) every moment no zone
may hold more than 3 of 5 fronts (2 of 3, 2 of 4) -- the alternating order of rounds 2-5 held all five in ONE zone for the 16-bit formats."""
import ctypes as C
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "endless-memory-gym_amd"))
from memory_gym_amd import _native  # noqa: E402

GRID = 14336
FRAME = {"u8": 21168, "f16": 42336, "f32": 84672}


def plan(n, frame_bytes, zones):
    z = (C.c_int * 256)()
    piece, lead = C.c_size_t(), C.c_size_t()
    k = _native.LIB.mg_obs_plan(n * frame_bytes, frame_bytes, zones, C.byref(piece), C.byref(lead), z, 256)
    assert 0 < k <= 256
    return list(z[:k]), piece.value, lead.value


def worst_share(n, frame_bytes, zones, alternate=False):
    """fraction of the launch during which one zone holds more than ceil(fronts / zones) + (zones == 2 and fronts odd ? 0 : 0) fronts, and the
    largest number of fronts one zone ever holds"""
    z, piece, lead = plan(n, frame_bytes, zones)
    if alternate:
        z = [p % zones for p in range(len(z))]
    bad = steps = most = 0
    for first in range(0, GRID, 64):  # the resident workgroups' first frame: every front advances with it
        fronts = [first + i * GRID for i in range((n - first + GRID - 1) // GRID) if first + i * GRID < n]
        where = [z[min((lead + f * frame_bytes) // piece, len(z) - 1)] for f in fronts]
        top = max(where.count(q) for q in range(zones))
        most = max(most, top)
        steps += 1
        bad += top > -(-len(fronts) // zones) + (1 if len(fronts) % zones == 0 else 0)  # more than an even split + 1 where the split is exact
    return bad / steps, most


@pytest.mark.parametrize("fmt", sorted(FRAME))
@pytest.mark.parametrize("n", [32768, 65536, 131072, 262144])
@pytest.mark.parametrize("zones", [2, 3])
def test_fronts_are_split_over_the_zones(fmt, n, zones):
    if n * FRAME[fmt] > 200 * 304 * (1 << 20):
        pytest.skip("more pieces than this test looks at")
    share, most = worst_share(n, FRAME[fmt], zones)
    fronts = -(-n // GRID)
    # (pieces straddle window boundaries, and a window is not a whole number of pieces: over many windows the two drift)
    assert share <= (0.12 if n <= 131072 and zones == 2 else 0.25), (fmt, n, zones, share, most)
    assert most <= -(-fronts // zones) + 2, (fmt, n, zones, most)
    if zones == 2:
        assert share <= worst_share(n, FRAME[fmt], zones, alternate=True)[0]


def test_the_alternating_order_was_what_kept_the_16_bit_formats_in_one_zone():
    share, most = worst_share(65536, FRAME["f16"], 2, alternate=True)
    assert most == 5 and share > 0.5  # all five fronts in one zone for most of the launch: 0.72 of peak (profiles/r06_float_formats.md)
    share, most = worst_share(65536, FRAME["f16"], 2)
    assert most <= 3 and share <= 0.05


def test_uint8_buffers_keep_the_alternating_order():
    z, _, _ = plan(65536, FRAME["u8"], 2)
    assert z == [p % 2 for p in range(len(z))]
