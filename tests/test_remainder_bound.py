"""The float-reciprocal remainder of the pivot kernels' small path (csrc/pip_advance.h: rcp_low / umod_tiny_low) restated
in numpy float32: for 2 <= g < 2^20 and a < 2^20 the scaled reciprocal, whatever its last bit, gives the quotient or one
less, so one correction yields a mod g.  (The kernels themselves are held against the oracle by the -m gpu tests; this
pins the arithmetic argument, including a reciprocal that is off by one ulp either way.)"""
import numpy as np


def test_single_correction_remainder():
    c = np.float32(0.99999904632568359375)
    assert float(c) == 1 - 2.0 ** -20
    rng = np.random.default_rng(1)
    for trial in range(8):
        g = rng.integers(2, 1 << 20, size=400_000, dtype=np.int64)
        a = rng.integers(0, 1 << 20, size=g.size, dtype=np.int64)
        if trial % 4 == 0:
            a = (a // g) * g                                       # exact multiples
        if trial % 4 == 1:
            a = np.minimum((a // g) * g + g - 1, (1 << 20) - 1)     # just below a multiple
        if trial % 4 == 2:
            g = rng.integers(2, 64, size=g.size)                    # large quotients
        for ulp in (-1, 0, 1):
            r32 = np.float32(1) / g.astype(np.float32)
            if ulp:
                r32 = np.nextafter(r32, np.float32(np.inf if ulp > 0 else -np.inf))
            rgl = (r32 * c).astype(np.float32)
            q = (a.astype(np.float32) * rgl).astype(np.float32).astype(np.int64)   # v_cvt_u32_f32 truncates
            r = a - q * g
            assert (r >= 0).all() and (r < 2 * g).all()
            r = np.where(r >= g, r - g, r)
            assert (r == a % g).all()
