"""The lane-per-job path generator (csrc/mg_mystery.hip: lane_path) compares A* f-costs through integer keys
(g_cost << 17) + round(sqrt(d2) * 2^17) instead of the reference's doubles g_cost + sqrt(d2)
(/root/reference/memory_gym/pygame_assets.py:701-724, heuristic :726-736).  This checks, over every reachable
(g_cost, d2), that the keys order exactly like the doubles and tie exactly where the doubles tie."""
import math

import numpy as np


def test_integer_keys_order_like_the_doubles():
    d2s = sorted({x * x + y * y for x in range(7) for y in range(7)})  # squared distances on the 7x7 grid
    assert max(d2s) < 80                                                # the LDS table has 80 entries
    g = np.arange(0, 8 * 48 + 1)                                        # integers(1, 9) <= 8 per step, a simple path has <= 48 steps
    f, k = [], []
    for d2 in d2s:
        h = math.sqrt(d2)
        f.append(g.astype(np.float64) + h)                              # what the reference compares
        k.append((g.astype(np.int64) << 17) + int(round(h * (1 << 17))))
    f, k = np.concatenate(f), np.concatenate(k)
    assert k.max() < (1 << 26)                                          # key << 6 | node fits 32 bits
    order = np.argsort(f, kind="stable")
    fs, ks = f[order], k[order]
    assert np.all(ks[1:] >= ks[:-1])
    assert np.array_equal(fs[1:] == fs[:-1], ks[1:] == ks[:-1])
    gaps = np.diff(fs)
    assert gaps[gaps > 0].min() > 2.0 ** -16                            # distinct sums are far apart (2.5e-3)
