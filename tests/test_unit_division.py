"""The float output formats are value / 255 (float32, then rounded to half where asked).  The stream-out computes the
quotient as q0 = v * r, q = fma(fma(-q0, 255, v), r, q0) with r = float32(1 / 255) instead of an IEEE division
(csrc/mg_raster*.hpp: byte_to_unit).  Exact rational arithmetic, every byte value: the result IS the correctly rounded
float32 quotient, i.e. what `obs.float() / 255` gives."""
from fractions import Fraction

import numpy as np


def round_to_f32(fr):
    """nearest-even float32 of an exact Fraction (normal range)"""
    if fr == 0:
        return np.float32(0)
    sign = -1 if fr < 0 else 1
    a = abs(fr)
    e = 0
    while Fraction(2) ** e > a:
        e -= 1
    while Fraction(2) ** (e + 1) <= a:
        e += 1
    ulp = Fraction(2) ** (e - 23)
    m = a / ulp
    n = m.numerator // m.denominator
    rem = m - n
    if rem > Fraction(1, 2) or (rem == Fraction(1, 2) and n % 2 == 1):
        n += 1
    return np.float32(sign * float(n * ulp))


def test_reciprocal_with_one_correction_equals_the_division():
    r = np.float32(1.0) / np.float32(255.0)
    R = Fraction(float(r))
    for v in range(256):
        want = np.float32(v) / np.float32(255.0)
        q0 = round_to_f32(Fraction(v) * R)
        rem = round_to_f32(Fraction(v) - Fraction(float(q0)) * 255)          # fma(-q0, 255, v): one rounding
        q = round_to_f32(Fraction(float(rem)) * R + Fraction(float(q0)))     # fma(rem, r, q0): one rounding
        assert q == want, v


def test_half_formats_need_no_correction():
    """The 16-bit output formats round the quotient once more (bfloat16: nearest even on the float32 bits; float16: IEEE).  For every byte
    the UNCORRECTED product v * r rounds to the same 16-bit value as the correctly rounded float32 quotient, so the stream-out of those
    formats multiplies once (csrc/mg_stream_out.hpp) -- although v * r itself differs from the quotient in its last bit for 126 bytes."""
    r = np.float32(1.0) / np.float32(255.0)
    v = np.arange(256, dtype=np.float32)
    want = v / np.float32(255.0)
    q0 = v * r
    assert (q0 != want).sum() > 100  # (the correction is not idle for float32)

    def bf16(x):
        u = x.view(np.uint32)
        return ((u + 0x7FFF + ((u >> 16) & 1)) >> 16).astype(np.uint16)

    assert np.array_equal(bf16(q0), bf16(want))
    assert np.array_equal(q0.astype(np.float16).view(np.uint16), want.astype(np.float16).view(np.uint16))
