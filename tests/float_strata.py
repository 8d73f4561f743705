"""Float operands over the whole encoding space, shared by the GPU parity tests and the CPU pin of the oracle.
TEST INFRASTRUCTURE."""
import numpy as np


def float_strata(rng, dt, n):
    """Float operands that cover the encodings a normal * 1e6 draw never reaches: -> (a, b), each
    the concatenation of
      * uniform random BIT PATTERNS (all 2^32 / 2^64 encodings: every exponent, denormals, infinities, NaN payloads);
      * near-denormal: exponent field 0..2 on both sides (denormal inputs, results that round into / out of the range);
      * boundary products / quotients: |a| tiny, |b| moderate (mul and div land around the smallest normal), and both
        huge (mul overflows to +-inf, div of huge by tiny too);
      * a huge exponent gap: |a| near the top of the range, |b| near the bottom (fmod walks hundreds of binades);
      * equal / negated / zero pairs (compare ties, -0.0 vs +0.0, x % x, x - x)."""
    f = np.dtype(dt.np_dtype)
    ut = {4: np.uint32, 8: np.uint64}[f.itemsize]
    mbits, ebits = (23, 8) if f.itemsize == 4 else (52, 11)
    emax = (1 << ebits) - 1
    k = max(n // 6, 1)

    def make(exp_lo, exp_hi, m):  # random sign and mantissa, exponent FIELD uniform in [exp_lo, exp_hi]
        sign = rng.integers(0, 2, m).astype(ut) << ut(f.itemsize * 8 - 1)
        expo = rng.integers(exp_lo, exp_hi + 1, m).astype(ut) << ut(mbits)
        mant = rng.integers(0, 1 << mbits, m, dtype=np.uint64).astype(ut)
        return (sign | expo | mant).view(f)

    def bits(m):
        return rng.integers(0, 1 << (f.itemsize * 8), m, dtype=np.uint64, endpoint=False).astype(ut).view(f) if f.itemsize == 4 else \
            rng.integers(0, np.iinfo(np.uint64).max, m, dtype=np.uint64, endpoint=True).view(f)

    bias = emax // 2
    a = [bits(k), make(0, 2, k), make(0, 3, k), make(emax - 3, emax - 1, k), make(emax - 40, emax - 1, k)]
    b = [bits(k), make(0, 2, k), make(bias - 2, bias + 2, k), make(emax - 3, emax - 1, k), make(0, 40, k)]
    # ties and signed zeros
    t = bits(k)
    t2 = t.copy()
    flip = rng.random(k) < 0.3
    t2.view(ut)[flip] ^= ut(1) << ut(f.itemsize * 8 - 1)
    z = rng.random(k) < 0.1
    t[z] = 0.0
    t2[z] = np.where(rng.random(int(z.sum())) < 0.5, -0.0, 0.0)
    a.append(t)
    b.append(t2)
    a, b = np.concatenate(a), np.concatenate(b)
    # +-inf and NaNs of both signs with random payloads (quiet and signalling), sprinkled over every stratum
    m = len(a)
    top = ut(1) << ut(f.itemsize * 8 - 1)
    for x in (a, b):
        x[rng.random(m) < 0.01] = np.inf
        x[rng.random(m) < 0.01] = -np.inf
        kk = rng.random(m) < 0.015
        payload = rng.integers(1, 1 << mbits, m, dtype=np.uint64).astype(ut)
        sign = np.where(rng.random(m) < 0.5, top, ut(0)).astype(ut)
        x.view(ut)[kk] = ((ut(emax) << ut(mbits)) | payload | sign)[kk]
    p = rng.permutation(m)
    return np.ascontiguousarray(a[p]), np.ascontiguousarray(b[p])
