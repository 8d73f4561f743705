// parse_num.hpp — text -> number for the Utf8 / LargeUtf8 -> numeric casts (cast_parse.hip).
//
// Reference: arrow-cast/src/cast/string.rs:66-120 `parse_string` -> `Parser::parse`
// (arrow-cast/src/parse.rs:446-528):
//   * integers: `parser_primitive!` :492-516 — trailing ASCII whitespace is dropped unless the text already ends in
//     a digit, the text must then END in a digit, and `atoi::FromRadix10SignedChecked` (atoi 3.1.0, Cargo.lock:601)
//     must consume ALL of it; if not, leading ASCII whitespace is dropped and it is tried once more.  atoi's
//     grammar: one optional `+` / `-`, then decimal digits accumulated with checked multiply / add (subtract for a
//     negative sign), so "-0" parses for unsigned types and any overflow is a failure.
//   * floats: `lexical_core::parse` (lexical-core 1.0.6, Cargo.lock:1959; default "standard" format) on the text,
//     and once more on the text with ASCII whitespace trimmed from both ends (:468-475).  Grammar restated from
//     lexical's documentation: optional sign; digits with an optional `.` (at least one digit on either side);
//     optional exponent `e|E [sign] digits` (digits required); or, case-insensitively, `nan`, `inf`, `infinity`
//     (after the optional sign); nothing may follow.  The value is the correctly rounded (nearest, ties to even)
//     binary float — unique, so any correct algorithm agrees bit for bit.  lexical's sources are not under
//     /root/reference: beyond the reference's own tests (parse.rs:2882-2955, cast/mod.rs:4879-4965) the grammar
//     is pinned only by that documentation.
//
// MI355X design: one lane parses one row with no memory traffic inside the loop (rows of <= 32 bytes live in four
// registers), and floats are converted with the Eisel-Lemire product (one or two 64x64->128 multiplies against a
// 651-entry table of 128-bit powers of five, pow5_128.inc) — no branches on the value, no fallback for up to 19
// significant digits (Mushtak & Lemire 2023).  Longer inputs are decided by the same product on the truncated
// significand w and on w + 1; only when those disagree does the row go to the exact big-integer comparison below
// (pn_slow_round_up), which a second, rarely launched kernel runs.
//
// The functions are plain C++ so that tests/cpp/parse_num_host_test.cpp can run them on the host against strtod /
// strtof over millions of inputs before they ever reach a GPU.
#pragma once
#include <cstdint>

#ifdef __HIPCC__
#define PN_FN __device__ __forceinline__
#define PN_SLOW_FN __device__ __noinline__
#define PN_CONST __device__ const
#else
#define PN_FN static inline
#define PN_SLOW_FN static
#define PN_CONST static const
#endif

PN_CONST uint64_t PN_POW5_128[651 * 2] = {
#include "pow5_128.inc"
};

PN_FN uint64_t pn_mul64(uint64_t a, uint64_t b, uint64_t* hi) {
#ifdef __HIPCC__
  *hi = __umul64hi(a, b);
  return a * b;
#else
  const unsigned __int128 p = (unsigned __int128)a * b;
  *hi = (uint64_t)(p >> 64);
  return (uint64_t)p;
#endif
}
PN_FN int pn_clz64(uint64_t v) {
#ifdef __HIPCC__
  return __clzll((long long)v);
#else
  return __builtin_clzll(v);
#endif
}

PN_FN bool pn_is_digit(uint8_t c) { return (uint8_t)(c - '0') < 10; }
// u8::is_ascii_whitespace: space, \t, \n, form feed, \r (NOT vertical tab)
PN_FN bool pn_is_space(uint8_t c) { return c == ' ' || c == '\t' || c == '\n' || c == 0x0C || c == '\r'; }

// ------------------------------------------------------------------------------------------------ integers
// `T` = the native type.  Returns false where the reference yields None.
template <typename T, typename B>
PN_FN bool pn_parse_int(const B& s, int64_t n, T* out) {
  constexpr bool is_signed = (T)-1 < (T)0;
  int64_t end = n;
  if (end == 0) return false;
  if (!pn_is_digit(s[end - 1])) {  // parse.rs:497-502
    while (end > 0 && pn_is_space(s[end - 1])) --end;
    if (end == 0 || !pn_is_digit(s[end - 1])) return false;
  }
  // atoi on [0, end); on failure once more without leading whitespace (:503-512).  The first attempt can only
  // succeed if there is no leading whitespace, so both attempts are "parse from the first non-blank byte".
  int64_t i = 0;
  while (i < end && pn_is_space(s[i])) ++i;
  bool neg = false;
  if (s[i] == '-') {
    neg = true;
    ++i;
  } else if (s[i] == '+') {
    ++i;
  }
  // magnitude with overflow tracking; the checked chain fails exactly when the signed value leaves T's range
  uint64_t mag = 0;
  bool ovf = false;
  for (; i < end; ++i) {
    const uint8_t c = s[i];
    if (!pn_is_digit(c)) return false;  // atoi stops here: not everything consumed
    const uint64_t d = (uint64_t)(c - '0');
    if (mag > (0xFFFFFFFFFFFFFFFFull - d) / 10) ovf = true;
    mag = mag * 10 + d;
  }
  if (ovf) return false;
  constexpr int bits = (int)sizeof(T) * 8;
  if (is_signed) {
    const uint64_t lim = (1ull << (bits - 1)) - (neg ? 0 : 1);  // |MIN| or MAX
    if (mag > lim) return false;
    *out = neg ? (T)(0 - mag) : (T)mag;
  } else {
    if (neg) {  // checked_sub from zero: only "-0…0" survives
      if (mag != 0) return false;
      *out = (T)0;
      return true;
    }
    if (bits < 64 && mag > ((1ull << (bits < 64 ? bits : 63)) - 1)) return false;
    *out = (T)mag;
  }
  return true;
}

// ------------------------------------------------------------------------------------------------ floats
template <typename F> struct PnFmt;
template <> struct PnFmt<double> {
  using bits_t = uint64_t;
  static constexpr int mant_bits = 52, min_exp = -1023, inf_power = 0x7FF;
  static constexpr int min_rte = -4, max_rte = 23;         // exponents where a decimal can be an exact tie
  static constexpr int smallest_q = -342, largest_q = 308;
};
template <> struct PnFmt<float> {
  using bits_t = uint32_t;
  static constexpr int mant_bits = 23, min_exp = -127, inf_power = 0xFF;
  static constexpr int min_rte = -17, max_rte = 10;
  static constexpr int smallest_q = -65, largest_q = 38;
};

struct PnBin {  // biased exponent + mantissa without the hidden bit; power2 == inf_power: infinity
  uint64_t mant;
  int32_t power2;
};

// w * 10^q -> nearest binary float (Eisel-Lemire).  Exact for every (w, q): no fallback.
template <typename F>
PN_FN PnBin pn_compute_float(int64_t q, uint64_t w) {
  using X = PnFmt<F>;
  PnBin r{0, 0};
  if (w == 0 || q < X::smallest_q) return r;
  if (q > X::largest_q) {
    r.power2 = X::inf_power;
    return r;
  }
  const int lz = pn_clz64(w);
  w <<= lz;
  // 128-bit product approximation with mant_bits + 3 bits of precision
  const int idx = 2 * (int)(q + 342);
  uint64_t hi, lo = pn_mul64(w, PN_POW5_128[idx], &hi);
  const uint64_t mask = 0xFFFFFFFFFFFFFFFFull >> (X::mant_bits + 3);
  if ((hi & mask) == mask) {
    uint64_t hi2;
    pn_mul64(w, PN_POW5_128[idx + 1], &hi2);
    lo += hi2;
    if (hi2 > lo) ++hi;
  }
  const int upperbit = (int)(hi >> 63);
  const int shift = upperbit + 64 - X::mant_bits - 3;
  uint64_t mant = hi >> shift;
  // floor(log2(10^q)) + 63, via 217706 / 2^16 ~ log2(10)
  int32_t power2 = (int32_t)(((217706 * q) >> 16) + 63) + upperbit - lz - X::min_exp;
  if (power2 <= 0) {  // subnormal
    if (-power2 + 1 >= 64) return r;
    mant >>= -power2 + 1;
    mant += mant & 1;
    mant >>= 1;
    r.mant = mant;
    r.power2 = mant < (1ull << X::mant_bits) ? 0 : 1;
    if (r.power2) r.mant &= ~(1ull << X::mant_bits);
    return r;
  }
  // exact tie: round to even
  if (lo <= 1 && q >= X::min_rte && q <= X::max_rte && (mant & 3) == 1 && (mant << shift) == hi) mant &= ~1ull;
  mant += mant & 1;
  mant >>= 1;
  if (mant >= (2ull << X::mant_bits)) {
    mant = 1ull << X::mant_bits;
    ++power2;
  }
  mant &= ~(1ull << X::mant_bits);
  if (power2 >= X::inf_power) {
    r.power2 = X::inf_power;
    return r;
  }
  r.mant = mant;
  r.power2 = power2;
  return r;
}

template <typename F>
PN_FN typename PnFmt<F>::bits_t pn_bits(bool neg, PnBin b) {
  using X = PnFmt<F>;
  using U = typename X::bits_t;
  return (U)(((U)neg << (sizeof(U) * 8 - 1)) | ((U)b.power2 << X::mant_bits) | (U)b.mant);
}

struct PnDec {
  uint64_t w;    // up to 19 significant digits
  int32_t q;     // value = w * 10^q (+ something in [0, 10^q) when inexact)
  uint8_t neg;
  uint8_t inexact;  // a nonzero digit was dropped
  uint8_t kind;     // 0 number, 1 infinity, 2 NaN
};

PN_FN uint8_t pn_lower(uint8_t c) { return (uint8_t)(c | 0x20); }

// case-insensitive match of s[i..n) against nan / inf / infinity
template <typename B>
PN_FN int pn_special(const B& s, int64_t i, int64_t n) {
  const int64_t m = n - i;
  if (m == 3) {
    const uint8_t a = pn_lower(s[i]), b = pn_lower(s[i + 1]), c = pn_lower(s[i + 2]);
    if (a == 'n' && b == 'a' && c == 'n') return 2;
    if (a == 'i' && b == 'n' && c == 'f') return 1;
    return 0;
  }
  if (m == 8) {
    const char* t = "infinity";
    for (int k = 0; k < 8; ++k)
      if (pn_lower(s[i + k]) != (uint8_t)t[k]) return 0;
    return 1;
  }
  return 0;
}

// one attempt on s[b..e): lexical_core::parse
template <typename B>
PN_FN bool pn_scan_float_range(const B& s, int64_t b, int64_t e, PnDec* d) {
  int64_t i = b;
  if (i >= e) return false;
  d->neg = 0;
  d->inexact = 0;
  d->kind = 0;
  if (s[i] == '-') {
    d->neg = 1;
    ++i;
  } else if (s[i] == '+') {
    ++i;
  }
  const int64_t after_sign = i;
  uint64_t w = 0;
  int nd = 0;           // significant digits in w
  int64_t dropped = 0;  // integer digits not folded into w
  int64_t frac_used = 0;
  int64_t ndigits = 0;
  for (; i < e && pn_is_digit(s[i]); ++i) {
    const uint64_t c = (uint64_t)(s[i] - '0');
    ++ndigits;
    if (nd < 19) {
      if (w != 0 || c != 0) {
        w = w * 10 + c;
        ++nd;
      }
    } else {
      ++dropped;
      if (c) d->inexact = 1;
    }
  }
  if (i < e && s[i] == '.') {
    ++i;
    for (; i < e && pn_is_digit(s[i]); ++i) {
      const uint64_t c = (uint64_t)(s[i] - '0');
      ++ndigits;
      if (nd < 19) {
        ++frac_used;
        if (w != 0 || c != 0) {
          w = w * 10 + c;
          ++nd;
        }
      } else if (c) {
        d->inexact = 1;
      }
    }
  }
  if (ndigits == 0) {  // not a number: the special values, whole remainder
    const int k = pn_special(s, after_sign, e);
    if (!k) return false;
    d->kind = (uint8_t)k;
    return true;
  }
  int64_t ex = 0;
  if (i < e && pn_lower(s[i]) == 'e') {
    ++i;
    bool eneg = false;
    if (i < e && (s[i] == '-' || s[i] == '+')) {
      eneg = s[i] == '-';
      ++i;
    }
    if (i >= e || !pn_is_digit(s[i])) return false;  // exponent digits are required
    for (; i < e && pn_is_digit(s[i]); ++i)
      if (ex < 100000000) ex = ex * 10 + (s[i] - '0');
    if (eneg) ex = -ex;
  }
  if (i != e) return false;
  int64_t q = ex + dropped - frac_used;
  if (q > 100000) q = 100000;
  if (q < -100000) q = -100000;
  d->w = w;
  d->q = (int32_t)q;
  return true;
}

// Float64Type::parse / Float32Type::parse (parse.rs:458-475): the text, then the text trimmed on both sides
template <typename B>
PN_FN bool pn_scan_float(const B& s, int64_t n, PnDec* d, int64_t* b_out, int64_t* e_out) {
  *b_out = 0;
  *e_out = n;
  if (pn_scan_float_range(s, 0, n, d)) return true;
  int64_t b = 0, e = n;
  while (b < e && pn_is_space(s[b])) ++b;
  while (e > b && pn_is_space(s[e - 1])) --e;
  if (b == 0 && e == n) return false;
  *b_out = b;
  *e_out = e;
  return pn_scan_float_range(s, b, e, d);
}

// Result of the fast path.  `need_slow`: the truncated significand straddles a rounding boundary — `bits` then
// holds the LOWER candidate (magnitude from w), and pn_slow_round_up decides whether to step to the next float.
template <typename F>
PN_FN typename PnFmt<F>::bits_t pn_convert(const PnDec& d, bool* need_slow) {
  using X = PnFmt<F>;
  using U = typename X::bits_t;
  *need_slow = false;
  if (d.kind == 1) return pn_bits<F>(d.neg, PnBin{0, X::inf_power});
  if (d.kind == 2) return pn_bits<F>(d.neg, PnBin{1ull << (X::mant_bits - 1), X::inf_power});  // quiet NaN
  const PnBin lo = pn_compute_float<F>(d.q, d.w);
  if (d.inexact) {
    const PnBin hi = pn_compute_float<F>(d.q, d.w + 1);
    if (hi.mant != lo.mant || hi.power2 != lo.power2) *need_slow = true;
  }
  return (U)pn_bits<F>(d.neg, lo);
}

// ------------------------------------------------------------------------------------------------ exact slow path
// x = D * 10^k (D = every significant digit, capped at PN_MAX_DIGITS with a sticky flag) against the midpoint
// m = (2M + 1) * 2^(E - 1) of the lower candidate M * 2^E and its successor.  Returns true when x rounds UP
// (x > m, or x == m and M is odd).
constexpr int PN_LIMBS = 136;       // 4352 bits: D (<= 2552 bits) * 2^1075 and 5^1110 * 2^54 * … both fit
constexpr int PN_MAX_DIGITS = 768;  // more can never change the rounding of a double (767 digits decide any tie)

struct PnBig {
  uint32_t d[PN_LIMBS];
  int n;
};

PN_SLOW_FN void pn_big_mul_small(PnBig* a, uint32_t m, uint32_t add) {
  uint64_t carry = add;
  for (int i = 0; i < a->n; ++i) {
    const uint64_t t = (uint64_t)a->d[i] * m + carry;
    a->d[i] = (uint32_t)t;
    carry = t >> 32;
  }
  if (carry && a->n < PN_LIMBS) a->d[a->n++] = (uint32_t)carry;
}
PN_SLOW_FN void pn_big_mul_pow5(PnBig* a, int64_t e) {
  while (e >= 13) {
    pn_big_mul_small(a, 1220703125u, 0);  // 5^13
    e -= 13;
  }
  uint32_t m = 1;
  for (; e > 0; --e) m *= 5;
  if (m != 1) pn_big_mul_small(a, m, 0);
}
PN_SLOW_FN void pn_big_shl(PnBig* a, int64_t s) {
  if (a->n == 0 || s <= 0) return;
  const int ws = (int)(s >> 5), bs = (int)(s & 31);
  int nn = a->n + ws + (bs ? 1 : 0);
  if (nn > PN_LIMBS) nn = PN_LIMBS;
  for (int i = nn - 1; i >= 0; --i) {
    const int src = i - ws;
    uint32_t v = 0;
    if (src >= 0 && src < a->n) v = a->d[src] << bs;
    if (bs && src - 1 >= 0 && src - 1 < a->n) v |= a->d[src - 1] >> (32 - bs);
    a->d[i] = v;
  }
  a->n = nn;
  while (a->n > 0 && a->d[a->n - 1] == 0) --a->n;
}
PN_SLOW_FN int pn_big_cmp(const PnBig* a, const PnBig* b) {
  if (a->n != b->n) return a->n < b->n ? -1 : 1;
  for (int i = a->n - 1; i >= 0; --i)
    if (a->d[i] != b->d[i]) return a->d[i] < b->d[i] ? -1 : 1;
  return 0;
}

// s[b..e) is a number already accepted by pn_scan_float_range; (mant, power2) the lower candidate
template <typename F, typename B>
PN_SLOW_FN bool pn_slow_round_up(const B& s, int64_t b, int64_t e, PnBin lo, PnBig* lhs, PnBig* rhs) {
  using X = PnFmt<F>;
  int64_t i = b;
  if (s[i] == '-' || s[i] == '+') ++i;
  // D and its decimal exponent
  lhs->n = 0;
  int nd = 0;
  int64_t k = 0;
  bool sticky = false, seen_point = false, started = false;
  uint32_t chunk = 0, chunk_mul = 1;
  for (; i < e; ++i) {
    const uint8_t c = s[i];
    if (c == '.') {
      seen_point = true;
      continue;
    }
    if (!pn_is_digit(c)) break;  // the exponent marker
    const uint32_t dg = (uint32_t)(c - '0');
    if (!started && dg == 0) {
      if (seen_point) --k;
      continue;
    }
    started = true;
    if (nd < PN_MAX_DIGITS) {
      chunk = chunk * 10 + dg;
      chunk_mul *= 10;
      ++nd;
      if (seen_point) --k;
      if (chunk_mul == 1000000000u) {
        pn_big_mul_small(lhs, chunk_mul, chunk);  // an empty D just takes the chunk
        chunk = 0;
        chunk_mul = 1;
      }
    } else {
      if (dg) sticky = true;
      if (!seen_point) ++k;
    }
  }
  if (chunk_mul != 1) pn_big_mul_small(lhs, chunk_mul, chunk);
  int64_t ex = 0;
  if (i < e) {  // 'e' / 'E'
    ++i;
    bool eneg = false;
    if (s[i] == '-' || s[i] == '+') {
      eneg = s[i] == '-';
      ++i;
    }
    for (; i < e; ++i)
      if (ex < 100000000) ex = ex * 10 + (s[i] - '0');
    if (eneg) ex = -ex;
  }
  k += ex;
  // midpoint (2M + 1) * 2^(E - 1)
  const uint64_t M = lo.power2 == 0 ? lo.mant : (lo.mant | (1ull << X::mant_bits));
  const int64_t E = (lo.power2 == 0 ? 1 : lo.power2) + X::min_exp - X::mant_bits;  // value = M * 2^E
  const uint64_t Mm = 2 * M + 1;
  const int64_t Em = E - 1;
  rhs->n = 0;
  rhs->d[rhs->n++] = (uint32_t)Mm;
  if (Mm >> 32) rhs->d[rhs->n++] = (uint32_t)(Mm >> 32);
  // compare D * 10^k with Mm * 2^Em over the integers
  if (k >= 0) {
    pn_big_mul_pow5(lhs, k);
    if (k >= Em) pn_big_shl(lhs, k - Em);
    else pn_big_shl(rhs, Em - k);
  } else {
    pn_big_mul_pow5(rhs, -k);
    if (Em - k >= 0) pn_big_shl(rhs, Em - k);
    else pn_big_shl(lhs, k - Em);
  }
  const int c = pn_big_cmp(lhs, rhs);
  if (c != 0) return c > 0;
  if (sticky) return true;
  return (M & 1) != 0;  // exact tie: to even
}

// next float up in magnitude (mantissa carry walks into the exponent; MAX + 1 ulp = infinity's encoding)
template <typename F>
PN_FN typename PnFmt<F>::bits_t pn_next_up_magnitude(typename PnFmt<F>::bits_t bits) {
  return (typename PnFmt<F>::bits_t)(bits + 1);
}
