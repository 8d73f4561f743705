// big_int_digits.hpp — shortest round-trip digits of integer-valued doubles in [2^53, 2^64), exact 64-bit arithmetic.
// Host + device: cast_string.hip uses it on the GPU, tests/cpp/big_int_digits_host_test.cpp checks it on the CPU against
// libstdc++'s shortest `to_chars` (tests/test_oracle_golden.py, no GPU needed).
#pragma once
#include <cstdint>

#if defined(__HIPCC__)
#define AH_BIGINT_HD __host__ __device__ __forceinline__
#else
#define AH_BIGINT_HD inline
#endif

// Integer-valued doubles with 2^53 <= |f| < 2^64 and a non-zero mantissa field (round 6) — what `Int64 as f64` produces
// beyond 2^53, the cast chain's general rows.  The value V = m2 << s and the half-ulp h = 2^(s-1) are exact 64-bit
// integers, so Ryu's interval [V - h, V + h] needs no 128-bit multiplication by a tabulated power of 5: this is d2d's
// "general" digit-removal loop (d2s.c step 4) started from the exact values with q = 0 digits removed up front — same
// trailing-zero bookkeeping, same tie and boundary rules, hence the same digits (Ryu's step 3 is an exact computation of
// floor(v / 10^q) for the same three numbers; its vr / vm trailing-zero flags can only differ from the exact ones when
// 5 divides none of mv, mv - 2, mv + 2 in a way that would matter, which needs h to be a multiple of 5 — it is a power
// of two).  ~250 instructions where d2d runs ~1 800; returns false for everything else (incl. exact powers of two, whose
// lower half-interval is h / 2).
AH_BIGINT_HD bool ah_big_int_shortest(uint64_t ieeeMantissa, uint32_t ieeeExponent, uint64_t* mantissa_out, int32_t* exponent_out) {
  const uint32_t s = ieeeExponent - 1075u;
  if (s - 1u > 10u || ieeeMantissa == 0) return false;  // s in 1..11: V < 2^64
  const uint64_t m2 = (1ull << 52) | ieeeMantissa;
  const uint64_t V = m2 << s, h = 1ull << (s - 1);
  const bool acceptBounds = (ieeeMantissa & 1) == 0;
  uint64_t vr = V, vm = V - h, vp = V + h - (acceptBounds ? 0u : 1u);
  bool vmIsTrailingZeros = acceptBounds, vrIsTrailingZeros = true;
  uint32_t lastRemovedDigit = 0;
  int32_t removed = 0;
  for (;;) {
    const uint64_t vp10 = vp / 10, vm10 = vm / 10;
    if (vp10 <= vm10) break;
    const uint64_t vr10 = vr / 10;
    vmIsTrailingZeros = vmIsTrailingZeros && (vm - vm10 * 10) == 0;
    vrIsTrailingZeros = vrIsTrailingZeros && lastRemovedDigit == 0;
    lastRemovedDigit = (uint32_t)(vr - vr10 * 10);
    vr = vr10, vp = vp10, vm = vm10;
    ++removed;
  }
  if (vmIsTrailingZeros) {
    for (;;) {
      const uint64_t vm10 = vm / 10;
      if (vm - vm10 * 10 != 0) break;
      const uint64_t vr10 = vr / 10;
      vrIsTrailingZeros = vrIsTrailingZeros && lastRemovedDigit == 0;
      lastRemovedDigit = (uint32_t)(vr - vr10 * 10);
      vr = vr10, vp /= 10, vm = vm10;
      ++removed;
    }
  }
  if (vrIsTrailingZeros && lastRemovedDigit == 5 && (vr & 1) == 0) lastRemovedDigit = 4;  // round half to even
  *mantissa_out = vr + (((vr == vm && (!acceptBounds || !vmIsTrailingZeros)) || lastRemovedDigit >= 5) ? 1u : 0u);
  *exponent_out = removed;
  return true;
}

