"""R-compatible uniform random numbers: `set.seed(s); runif(n, a, b)` of R's
default generator (Mersenne-Twister, "Inversion", as in R >= 3.6), so that the random-walk initial
positions of `euclidean_embedding()` (reference R/core.R:407-415) can be reproduced exactly for the
seed an R user would pass to set.seed().

Algorithm (R sources src/main/RNG.c, public): the seed is scrambled 50 times with the LCG
x <- 69069 x + 1 (mod 2^32), the next 625 LCG outputs fill the state (word 0 is then replaced by the
position counter 624), MT19937 generates 32-bit words, a draw is word * 2^-32 nudged into (0,1)."""
from __future__ import annotations

import numpy as np

_N, _M = 624, 397
_MATRIX_A, _UPPER, _LOWER = 0x9908B0DF, 0x80000000, 0x7FFFFFFF
_I2_32M1 = 2.328306437080797e-10  # 1/(2^32 - 1), R's fixup() constant


class RUnif:
    def __init__(self, seed: int):
        s = int(seed) & 0xFFFFFFFF
        for _ in range(50):
            s = (69069 * s + 1) & 0xFFFFFFFF
        state = []
        for _ in range(_N + 1):
            s = (69069 * s + 1) & 0xFFFFFFFF
            state.append(s)
        self.mt = state[1:]          # word 0 of R's i_seed is the position counter
        self.mti = _N

    def _refill(self):
        mt = self.mt
        for kk in range(_N - _M):
            y = (mt[kk] & _UPPER) | (mt[kk + 1] & _LOWER)
            mt[kk] = mt[kk + _M] ^ (y >> 1) ^ (_MATRIX_A if y & 1 else 0)
        for kk in range(_N - _M, _N - 1):
            y = (mt[kk] & _UPPER) | (mt[kk + 1] & _LOWER)
            mt[kk] = mt[kk + (_M - _N)] ^ (y >> 1) ^ (_MATRIX_A if y & 1 else 0)
        y = (mt[_N - 1] & _UPPER) | (mt[0] & _LOWER)
        mt[_N - 1] = mt[_M - 1] ^ (y >> 1) ^ (_MATRIX_A if y & 1 else 0)
        self.mti = 0

    def unif_rand(self) -> float:
        if self.mti >= _N:
            self._refill()
        y = self.mt[self.mti]
        self.mti += 1
        y ^= y >> 11
        y ^= (y << 7) & 0x9D2C5680
        y ^= (y << 15) & 0xEFC60000
        y ^= y >> 18
        v = (y & 0xFFFFFFFF) * 2.3283064365386963e-10
        if v <= 0.0:
            return 0.5 * _I2_32M1
        if 1.0 - v <= 0.0:
            return 1.0 - 0.5 * _I2_32M1
        return v

    def runif(self, n: int, lo: float = 0.0, hi: float = 1.0) -> np.ndarray:
        return np.array([lo + (hi - lo) * self.unif_rand() for _ in range(n)], dtype=np.float64)

    # numpy.random.Generator look-alike, enough for core.prepare_layout_call
    def uniform(self, lo, hi, size):
        count = int(np.prod(size))
        return self.runif(count, lo, hi).reshape(size)

    def integers(self, lo, hi=None):
        """One integer in [lo, hi) from a single draw (what the R shim does for the native seed:
        `unif_rand() * 2^53`)."""
        if hi is None:
            lo, hi = 0, lo
        return int(lo) + int(self.unif_rand() * 9007199254740992.0) % max(1, int(hi) - int(lo))
