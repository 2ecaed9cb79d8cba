"""Philox4x32-10 counter-based uniforms -- TEST INFRASTRUCTURE (oracle side).

The reference draws from CPython's process-global Mersenne Twister through
``random.uniform`` (reference lib4mc/ProbLib.py:41).  The MI355X sampler cannot
share one sequential generator across 10^5 lanes, so the build defines a
counter-based stream instead: sample ``i``, draw ``k`` -> one 53-bit uniform that
depends only on ``(seed, i, k)``.  This file is the plain-integer statement of
that stream; the HIP kernel (montecarlo_amd/csrc) and oracle/mc_oracle.c must
produce the same bits, and tests inject exactly these uniforms into the
reference sampler by patching ``random.uniform`` (tests/golden/make_golden.py).

Mapping (fixed, part of the ABI -- see include/mi355mc.h):
  key     = (seed & 0xffffffff, seed >> 32)
  counter = (i & 0xffffffff, i >> 32, k >> 1, 0)
  words   = philox4x32_10(counter, key)            -> w0..w3
  (a, b)  = (w0, w1) if k even else (w2, w3)
  r       = ((a >> 5) * 2**26 + (b >> 6)) / 2**53  (the genrand_res53 recipe)
"""
from __future__ import annotations

M0 = 0xD2511F53
M1 = 0xCD9E8D57
W0 = 0x9E3779B9
W1 = 0xBB67AE85
MASK = 0xFFFFFFFF


def philox4x32_10(ctr, key):
    c0, c1, c2, c3 = ctr
    k0, k1 = key
    for rnd in range(10):
        if rnd:
            k0 = (k0 + W0) & MASK
            k1 = (k1 + W1) & MASK
        p0 = M0 * c0
        p1 = M1 * c2
        c0, c1, c2, c3 = ((p1 >> 32) ^ c1 ^ k0) & MASK, p1 & MASK, ((p0 >> 32) ^ c3 ^ k1) & MASK, p0 & MASK
    return c0, c1, c2, c3


def uniform53(seed: int, sample_id: int, draw: int) -> float:
    w = philox4x32_10((sample_id & MASK, (sample_id >> 32) & MASK, (draw >> 1) & MASK, 0),
                      (seed & MASK, (seed >> 32) & MASK))
    a, b = (w[0], w[1]) if (draw & 1) == 0 else (w[2], w[3])
    return ((a >> 5) * 67108864 + (b >> 6)) / 9007199254740992.0


class SampleStream:
    """Callable uniform source for one sample id; counts the draws it served."""

    def __init__(self, seed: int, sample_id: int):
        self.seed, self.sample_id, self.draws = seed, sample_id, 0

    def __call__(self) -> float:
        r = uniform53(self.seed, self.sample_id, self.draws)
        self.draws += 1
        return r
