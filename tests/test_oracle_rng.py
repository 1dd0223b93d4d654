"""Pin the oracle's RNG restatement (oracle/mgo_rng.h) against numpy's Generator(PCG64(SeedSequence(s)))."""
import numpy as np
import pytest

import oracle_lib


@pytest.mark.parametrize("seed", [0, 1, 2, 12345, 2**31 - 1, 2**32, 2**32 + 7, 123456789012, 2**63 - 1])
def test_mixed_draws_match_numpy(seed):
    L = oracle_lib.lib()
    n = 20000
    prng = np.random.default_rng(seed ^ 0xABCDEF)
    ops = prng.integers(0, 4, n).astype(np.int32)
    lo = prng.integers(-200, 200, n).astype(np.int64)
    span = prng.choice([1, 2, 3, 4, 5, 9, 25, 36, 100, 360, 7056, 2**31, 2**32 - 1], n).astype(np.int64)
    hi = lo + span
    g = np.random.Generator(np.random.PCG64(np.random.SeedSequence(seed)))
    exp = np.empty(n)
    for i in range(n):
        if ops[i] == 0:
            exp[i] = g.integers(lo[i], hi[i])
        elif ops[i] == 1:
            exp[i] = g.random()
        elif ops[i] == 2:
            exp[i] = int(g.bit_generator.random_raw()) >> 11
        else:
            exp[i] = g.uniform(lo[i] / 1e6, hi[i] / 1e6)
    out = np.empty(n)
    L.mgo_test_rng(seed, ops.ctypes.data, lo.ctypes.data, hi.ctypes.data, n, out.ctypes.data)
    assert np.array_equal(out, exp)


def test_known_answers_from_survey():
    # raw PCG64(SeedSequence(1)) first outputs and integers(0,25) x6
    g = np.random.Generator(np.random.PCG64(np.random.SeedSequence(1)))
    assert [int(g.integers(0, 25)) for _ in range(6)] == [11, 12, 18, 23, 0, 3]
    L = oracle_lib.lib()
    ops = np.zeros(6, np.int32)
    lo = np.zeros(6, np.int64)
    hi = np.full(6, 25, np.int64)
    out = np.empty(6)
    L.mgo_test_rng(1, ops.ctypes.data, lo.ctypes.data, hi.ctypes.data, 6, out.ctypes.data)
    assert out.tolist() == [11, 12, 18, 23, 0, 3]
