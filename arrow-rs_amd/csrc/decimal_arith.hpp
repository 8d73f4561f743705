// decimal_arith.hpp — the arithmetic and the type rules of Decimal128 add / sub / mul / div / rem (arith_decimal.hip).
//
// Reference: `decimal_op` (arrow-arith/src/numeric.rs:971-1103, "the Hive decimal arithmetic rules") over
// ArrowNativeTypeOp for i128 (arrow-array/src/arithmetic.rs:147-284: add_checked, sub_checked, mul_checked,
// div_checked, mod_checked, pow_checked and their error texts).  Every arm is
//     l.mul_checked(l_mul)? <op>_checked (r.mul_checked(r_mul)?)
// with l_mul / r_mul the powers of ten that align the two scales (1 when the scales agree), evaluated on valid slots
// only (try_op!).
//
// Plain C++ on purpose (like parse_num.hpp / temporal_cast.hpp): tests/cpp/decimal_arith_host_test.cpp compiles this
// header for the HOST and compares `dec_row` with the compiler's native __int128 arithmetic over edge and random
// operands before the same source reaches a GPU.  The device has no 128-bit multiply-with-overflow or division
// (no __muloti4 / __divti3 in the device runtime), so both are written out here on 64-bit limbs.
#pragma once
#include <cstdint>
#include <cstdio>
#include <string>

#include "../../include/arrow_hip.h"

#ifdef __HIPCC__
#define DA_FN __host__ __device__ __forceinline__
#else
#define DA_FN static inline
#endif

namespace da {

typedef __int128 i128;
typedef unsigned __int128 u128;

enum DOp : int { D_ADD = 0, D_SUB = 1, D_MUL = 2, D_DIV = 3, D_REM = 4 };
// which step of `l.mul_checked(l_mul)? op (r.mul_checked(r_mul)?)` failed
enum DFail : int { D_OK = 0, D_FAIL_L_SCALE = 1, D_FAIL_R_SCALE = 2, D_FAIL_OP = 3, D_FAIL_DIV_ZERO = 4 };

struct DParams {
  int op;
  int scaled;  // 0: both multipliers are one (the reference's equal-scale fast path for add / sub; always for mul)
  i128 l_mul, r_mul;
};

DA_FN u128 da_abs(i128 v) { return v < 0 ? (u128)0 - (u128)v : (u128)v; }

// i128::checked_mul on 64-bit limbs
DA_FN bool da_mul_checked(i128 a, i128 b, i128* out) {
  const bool neg = (a < 0) != (b < 0);
  const u128 ua = da_abs(a), ub = da_abs(b);
  const uint64_t a0 = (uint64_t)ua, a1 = (uint64_t)(ua >> 64), b0 = (uint64_t)ub, b1 = (uint64_t)(ub >> 64);
  if (a1 && b1) return false;
  const u128 cross = a1 ? (u128)a1 * b0 : (u128)b1 * a0;  // 64 x 64 -> 128
  if (cross >> 64) return false;
  const u128 low = (u128)a0 * b0;
  const u128 res = low + (cross << 64);
  if (res < low) return false;
  if (neg) {
    if (res > ((u128)1 << 127)) return false;
    *out = (i128)((u128)0 - res);
  } else {
    if (res >> 127) return false;
    *out = (i128)res;
  }
  return true;
}

DA_FN int da_clz128(u128 v) {  // v != 0
  const uint64_t hi = (uint64_t)(v >> 64), lo = (uint64_t)v;
  return hi ? __builtin_clzll(hi) : 64 + __builtin_clzll(lo);
}

// unsigned 128 / 128 -> quotient and remainder: restoring shift-subtract over the bits the quotient can have
DA_FN void da_udivrem(u128 n, u128 d, u128* q, u128* r) {  // d != 0
  if (d > n) {
    *q = 0;
    *r = n;
    return;
  }
  int shift = da_clz128(d) - da_clz128(n);  // >= 0
  d <<= shift;
  u128 quo = 0;
  for (; shift >= 0; --shift) {
    quo <<= 1;
    if (n >= d) {
      n -= d;
      quo |= 1;
    }
    d >>= 1;
  }
  *q = quo;
  *r = n;
}

// i128 `/` and `%` (truncating, remainder takes the dividend's sign); the caller excludes d == 0 and MIN / -1
DA_FN void da_divrem(i128 n, i128 d, i128* q, i128* r) {
  u128 uq, ur;
  da_udivrem(da_abs(n), da_abs(d), &uq, &ur);
  *q = ((n < 0) != (d < 0)) ? (i128)((u128)0 - uq) : (i128)uq;
  *r = (n < 0) ? (i128)((u128)0 - ur) : (i128)ur;
}

DA_FN i128 da_min() { return (i128)((u128)1 << 127); }

// one row; on failure `fail` says which step, and the scaled operands (as far as they were computed) are left in
// *ls / *rs so the host can print the reference's message
DA_FN int dec_row(const DParams& p, i128 l, i128 r, i128* out, i128* ls, i128* rs) {
  i128 a = l, b = r;
  if (p.scaled) {
    if (!da_mul_checked(l, p.l_mul, &a)) return D_FAIL_L_SCALE;
    *ls = a;
    if (!da_mul_checked(r, p.r_mul, &b)) return D_FAIL_R_SCALE;
  }
  *ls = a;
  *rs = b;
  switch (p.op) {
    case D_ADD: {
      i128 s = (i128)((u128)a + (u128)b);
      if (((a ^ s) & (b ^ s)) < 0) return D_FAIL_OP;
      *out = s;
      return D_OK;
    }
    case D_SUB: {
      i128 s = (i128)((u128)a - (u128)b);
      if (((a ^ b) & (a ^ s)) < 0) return D_FAIL_OP;
      *out = s;
      return D_OK;
    }
    case D_MUL: return da_mul_checked(a, b, out) ? D_OK : D_FAIL_OP;
    default: {
      if (b == 0) return D_FAIL_DIV_ZERO;
      if (a == da_min() && b == -1) return D_FAIL_OP;  // checked_div / checked_rem
      i128 q, rem;
      da_divrem(a, b, &q, &rem);
      *out = p.op == D_DIV ? q : rem;
      return D_OK;
    }
  }
}

// ------------------------------------------------------------------ type rules (host)
static inline std::string i128_text(i128 v) {  // {:?} of an i128
  if (v == 0) return "0";
  u128 u = da_abs(v);
  char buf[48];
  int n = 0;
  while (u) {
    buf[n++] = (char)('0' + (int)(u % 10));
    u /= 10;
  }
  std::string s = v < 0 ? "-" : "";
  while (n) s += buf[--n];
  return s;
}

static inline std::string decimal_type_text(const ah_data_type& t) {
  char b[48];
  snprintf(b, sizeof b, "Decimal128(%d, %d)", t.precision, t.scale);
  return b;
}

static inline bool pow10_checked(int exp, i128* out) {  // 10.pow_checked(exp): i128 holds 10^38
  if (exp < 0 || exp > 38) return false;
  i128 v = 1;
  for (int i = 0; i < exp; ++i) v *= 10;
  *out = v;
  return true;
}
static inline i128 pow10_wrapping(unsigned exp) {  // 10.pow_wrapping(exp)
  u128 v = 1;
  for (unsigned i = 0; i < exp && i < 200; ++i) v *= 10;  // 10^k is 0 mod 2^128 from k = 128 on
  return (i128)v;
}

constexpr int kMaxPrecision = 38, kMaxScale = 38;  // Decimal128Type::MAX_PRECISION / MAX_SCALE

struct DPlan {
  DParams p;
  ah_data_type result;
  ah_status status;  // AH_OK or the error raised BEFORE any row is evaluated (pow_checked, mul's output scale)
  std::string message;
  ah_status post_status;  // with_precision_and_scale(..)?: raised only after every row succeeded
  std::string post_message;
};

static inline int sat_i8(int v) { return v < -128 ? -128 : (v > 127 ? 127 : v); }
static inline int sat_u8(int v) { return v < 0 ? 0 : (v > 255 ? 255 : v); }

// decimal_op's match on `op` (numeric.rs:992-1100).  `op` is an ah_arith_op.
static inline DPlan make_decimal_plan(ah_arith_op op, const ah_data_type& lt, const ah_data_type& rt, const char* op_sym) {
  DPlan d{};
  d.status = AH_OK;
  d.post_status = AH_OK;
  d.result = lt;
  const int p1 = lt.precision, s1 = lt.scale, p2 = rt.precision, s2 = rt.scale;
  auto overflow_pow = [&](int exp) {
    d.status = AH_ARITHMETIC_OVERFLOW;
    d.message = "Overflow happened on: 10 ^ " + std::to_string(exp);
  };
  if (op == AH_ADD || op == AH_ADD_WRAPPING || op == AH_SUB || op == AH_SUB_WRAPPING) {
    const int rs = s1 > s2 ? s1 : s2;
    // (result_scale.saturating_add((p1 as i8 - s1).max(p2 as i8 - s2)) as u8).saturating_add(1).min(MAX_PRECISION)
    const int whole = ((int8_t)p1 - s1) > ((int8_t)p2 - s2) ? ((int8_t)p1 - s1) : ((int8_t)p2 - s2);
    int prec = (uint8_t)(int8_t)sat_i8(rs + (int)(int8_t)whole);
    prec = sat_u8(prec + 1);
    if (prec > kMaxPrecision) prec = kMaxPrecision;
    d.p.op = (op == AH_ADD || op == AH_ADD_WRAPPING) ? D_ADD : D_SUB;
    // `(result_scale - s1) as _` feeds a u32 exponent: the difference is never negative
    if (!pow10_checked(rs - s1, &d.p.l_mul)) return overflow_pow(rs - s1), d;
    if (!pow10_checked(rs - s2, &d.p.r_mul)) return overflow_pow(rs - s2), d;
    d.p.scaled = s1 != s2;
    d.result.precision = prec;
    d.result.scale = rs;
  } else if (op == AH_MUL || op == AH_MUL_WRAPPING) {
    int prec = sat_u8(p1 + sat_u8(p2 + 1));
    if (prec > kMaxPrecision) prec = kMaxPrecision;
    const int rs = sat_i8(s1 + s2);
    if (rs > kMaxScale) {
      d.status = AH_INVALID_ARGUMENT;
      d.message = "Output scale of " + decimal_type_text(lt) + " " + op_sym + " " + decimal_type_text(rt) +
                  " would exceed max scale of " + std::to_string(kMaxScale);
      return d;
    }
    d.p.op = D_MUL;
    d.p.scaled = 0;
    d.p.l_mul = d.p.r_mul = 1;
    d.result.precision = prec;
    d.result.scale = rs;
  } else if (op == AH_DIV) {
    int rs = sat_i8(s1 + 4);
    if (rs > kMaxScale) rs = kMaxScale;
    const int mul_pow = (int8_t)(rs - s1 + s2);  // i8 arithmetic
    int prec = (uint8_t)(int8_t)sat_i8(mul_pow + (int)(int8_t)p1);
    if (prec > kMaxPrecision) prec = kMaxPrecision;
    d.p.op = D_DIV;
    d.p.scaled = 1;
    d.p.l_mul = d.p.r_mul = 1;
    if (mul_pow > 0) {
      if (!pow10_checked(mul_pow, &d.p.l_mul)) return overflow_pow(mul_pow), d;
    } else if (mul_pow < 0) {
      const int e = (uint8_t)(int8_t)(-mul_pow);  // mul_pow.neg_wrapping() as u32
      if (!pow10_checked(e, &d.p.r_mul)) return overflow_pow(e), d;
    }
    d.result.precision = prec;
    d.result.scale = rs;
  } else {  // AH_REM
    const int rs = s1 > s2 ? s1 : s2;
    const int whole = ((int8_t)p1 - s1) < ((int8_t)p2 - s2) ? ((int8_t)p1 - s1) : ((int8_t)p2 - s2);
    int prec = (uint8_t)(int8_t)sat_i8(rs + (int)(int8_t)whole);
    if (prec > kMaxPrecision) prec = kMaxPrecision;
    d.p.op = D_REM;
    d.p.scaled = 1;
    d.p.l_mul = pow10_wrapping((unsigned)(rs - s1));
    d.p.r_mul = pow10_wrapping((unsigned)(rs - s2));
    d.result.precision = prec;
    d.result.scale = rs;
  }
  // with_precision_and_scale -> validate_decimal_precision_and_scale (arrow-array/src/types.rs:1397-1425)
  const int prec = d.result.precision, sc = d.result.scale;
  if (prec == 0) {
    d.post_status = AH_INVALID_ARGUMENT;
    d.post_message = "precision cannot be 0, has to be between [1, 38]";
  } else if (sc > kMaxScale) {
    d.post_status = AH_INVALID_ARGUMENT;
    d.post_message = "scale " + std::to_string(sc) + " is greater than max " + std::to_string(kMaxScale);
  } else if (sc > 0 && sc > prec) {
    d.post_status = AH_INVALID_ARGUMENT;
    d.post_message = "scale " + std::to_string(sc) + " is greater than precision " + std::to_string(prec);
  }
  return d;
}

// the reference's message for a failed row (ArrowNativeTypeOp's texts, arithmetic.rs:160-240)
static inline ah_status row_error(const DParams& p, int fail, i128 l, i128 r, i128 ls, i128 rs, std::string* msg) {
  static const char* sym[] = {"+", "-", "*", "/", "%"};
  switch (fail) {
    case D_FAIL_L_SCALE: *msg = "Overflow happened on: " + i128_text(l) + " * " + i128_text(p.l_mul); return AH_ARITHMETIC_OVERFLOW;
    case D_FAIL_R_SCALE: *msg = "Overflow happened on: " + i128_text(r) + " * " + i128_text(p.r_mul); return AH_ARITHMETIC_OVERFLOW;
    case D_FAIL_DIV_ZERO: *msg = "Divide by zero error"; return AH_DIVIDE_BY_ZERO;
    default: *msg = "Overflow happened on: " + i128_text(ls) + " " + sym[p.op] + " " + i128_text(rs); return AH_ARITHMETIC_OVERFLOW;
  }
}


// ------------------------------------------------------------------ Decimal128 -> Decimal128 casts
// cast_decimal_to_decimal_same_type (arrow-cast/src/cast/decimal.rs:448-489) with make_upscaler (:161-198),
// make_downscaler (:210-267) and apply_decimal_cast (:351-377).
enum CMode : int { C_CLONE = 0, C_UP = 1, C_DOWN = 2, C_ZEROS = 3 };
struct CParams {
  int mode;
  int infallible;  // the `unary` path: every input fits, no precision test
  i128 k;          // 10^delta: multiplier (C_UP) or divisor (C_DOWN)
  i128 half;       // C_DOWN: k / 2
  i128 max_v;      // 10^output_precision - 1 (is_valid_decimal_precision)
};

// one row; false = the value cannot be represented (safe: null; unsafe: the first such row is the error).
// *stage: 1 = the rescale itself overflowed ("Overflowing on"), 2 = the precision test failed ("too large / too small")
DA_FN bool dcast_row(const CParams& p, i128 x, i128* out, int* stage) {
  i128 v = x;
  if (p.mode == C_UP) {
    if (p.infallible) {
      *out = (i128)((u128)x * (u128)p.k);  // mul_wrapping
      return true;
    }
    if (!da_mul_checked(x, p.k, &v)) {
      *stage = 1;
      return false;
    }
  } else if (p.mode == C_DOWN) {
    i128 d, r;
    da_divrem(x, p.k, &d, &r);  // k >= 10: div_wrapping / mod_wrapping cannot overflow
    if (x >= 0 ? r >= p.half : r <= -p.half) d += (x >= 0 ? 1 : -1);  // round half away from zero
    v = d;
    if (p.infallible) {
      *out = v;
      return true;
    }
  } else if (p.mode == C_ZEROS) {
    *out = 0;
    return true;
  }
  if (v > p.max_v || v < -p.max_v) {
    *stage = 2;
    return false;
  }
  *out = v;
  return true;
}

struct CPlan {
  CParams p;
  ah_status status;  // raised before any row (upscale beyond the table)
  std::string message;
  ah_status post_status;  // with_precision_and_scale(output)?
  std::string post_message;
};

static inline CPlan make_decimal_cast_plan(const ah_data_type& from, const ah_data_type& to) {
  CPlan c{};
  c.status = c.post_status = AH_OK;
  const int ip = from.precision, is = from.scale, op = to.precision, os = to.scale;
  i128 maxv = 0;
  pow10_checked(op >= 0 && op <= 38 ? op : 0, &maxv);
  c.p.max_v = maxv - 1;  // MAX_DECIMAL128_FOR_EACH_PRECISION[op]
  c.p.k = 1;
  if (is == os && ip <= op) {
    c.p.mode = C_CLONE;
  } else if (is <= os) {
    const int delta = (int8_t)(os - is);
    if (delta < 0 || delta > 38 || !pow10_checked(delta, &c.p.k)) {  // MAX_FOR_EACH_PRECISION.get(delta) is None
      c.status = AH_CAST_ERROR;
      c.message = "Cannot cast to Decimal128(" + std::to_string(op) + ", " + std::to_string(os) + "). Value overflows for output scale";
      return c;
    }
    c.p.mode = C_UP;
    c.p.infallible = ((int8_t)ip + delta) <= (int8_t)op;
  } else {
    const int delta = (int8_t)(is - os);
    if (delta < 0 || delta > 38 || !pow10_checked(delta, &c.p.k)) {
      c.p.mode = C_ZEROS;  // every value rounds to zero
      c.p.infallible = 1;
    } else {
      c.p.mode = C_DOWN;
      c.p.half = c.p.k / 2;
      c.p.infallible = ((int8_t)ip - delta) < (int8_t)op;
    }
  }
  if (op == 0) {
    c.post_status = AH_INVALID_ARGUMENT;
    c.post_message = "precision cannot be 0, has to be between [1, 38]";
  } else if (op > kMaxPrecision) {
    c.post_status = AH_INVALID_ARGUMENT;
    c.post_message = "precision " + std::to_string(op) + " is greater than max 38";
  } else if (os > kMaxScale) {
    c.post_status = AH_INVALID_ARGUMENT;
    c.post_message = "scale " + std::to_string(os) + " is greater than max 38";
  } else if (os > 0 && os > op) {
    c.post_status = AH_INVALID_ARGUMENT;
    c.post_message = "scale " + std::to_string(os) + " is greater than precision " + std::to_string(op);
  }
  return c;
}

// format_decimal_str_internal (arrow-data/src/decimal.rs:1137-1167)
static inline std::string format_decimal_str(const std::string& value_str, size_t precision, int scale, bool safe_decimal) {
  const bool neg = !value_str.empty() && value_str[0] == '-';
  const std::string sign = neg ? "-" : "", rest = neg ? value_str.substr(1) : value_str;
  const size_t bound = safe_decimal ? (precision < rest.size() ? precision : rest.size()) + sign.size() : value_str.size();
  const std::string v = value_str.substr(0, bound);
  if (scale == 0) return v;
  if (scale < 0) return v + std::string((size_t)(-scale), '0');
  if (rest.size() > (size_t)scale) return v.substr(0, v.size() - (size_t)scale) + "." + v.substr(v.size() - (size_t)scale);
  return sign + "0." + std::string((size_t)scale - rest.size(), '0') + rest;
}

// the unsafe-mode error of the first failing row (decimal.rs:330-349; validate_decimal128_precision, arrow-data/src/decimal.rs:1024-1061)
static inline ah_status dcast_row_error(const CParams& p, const ah_data_type& to, i128 x, std::string* msg) {
  i128 v = 0;
  int stage = 0;
  dcast_row(p, x, &v, &stage);
  if (stage == 1) {
    *msg = "Cannot cast to Decimal128(" + std::to_string(to.precision) + ", " + std::to_string(to.scale) + "). Overflowing on " + i128_text(x);
    return AH_CAST_ERROR;
  }
  // recompute the rescaled value that failed the precision test
  i128 r = x;
  if (p.mode == C_UP) da_mul_checked(x, p.k, &r);
  else if (p.mode == C_DOWN) {
    i128 d, rem;
    da_divrem(x, p.k, &d, &rem);
    if (x >= 0 ? rem >= p.half : rem <= -p.half) d += (x >= 0 ? 1 : -1);
    r = d;
  }
  const bool large = r > p.max_v;
  *msg = format_decimal_str(i128_text(r), (size_t)to.precision, to.scale, false) + (large ? " is too large" : " is too small") +
         " to store in a Decimal128 of precision " + std::to_string(to.precision) + (large ? ". Max is " : ". Min is ") +
         format_decimal_str(i128_text(large ? p.max_v : -p.max_v), (size_t)to.precision, to.scale, true);
  return AH_INVALID_ARGUMENT;
}


// ------------------------------------------------------------------ integer -> Decimal128 casts
// cast_integer_to_decimal (arrow-cast/src/cast/mod.rs:366-443).  scale >= 0: `v as i128` * 10^scale, checked, then the
// precision test; scale < 0: v / 10^|scale| IN THE SOURCE TYPE (truncating; a factor that does not fit the source type
// makes every result zero), then the precision test.
enum IMode : int { I_MUL = 0, I_DIV = 1, I_ZEROS = 2 };
struct IParams {
  int mode;
  i128 k;      // 10^|scale|
  i128 max_v;  // 10^precision - 1
};
// stage: 1 = the multiply overflowed, 2 = the precision test failed
DA_FN bool icast_row(const IParams& p, i128 x, i128* out, int* stage) {
  i128 v;
  if (p.mode == I_MUL) {
    if (!da_mul_checked(x, p.k, &v)) {
      *stage = 1;
      return false;
    }
  } else if (p.mode == I_DIV) {
    i128 r;
    da_divrem(x, p.k, &v, &r);
  } else {
    *out = 0;
    return true;
  }
  if (v > p.max_v || v < -p.max_v) {
    *stage = 2;
    return false;
  }
  *out = v;
  return true;
}

struct IPlan {
  IParams p;
  ah_status status;
  std::string message;
  ah_status post_status;
  std::string post_message;
};
// `src_max` = the source type's MAX (decides whether 10^|scale| exists in the source type)
static inline IPlan make_int_to_decimal_plan(u128 src_max, const ah_data_type& to) {
  IPlan c{};
  c.status = c.post_status = AH_OK;
  const int op = to.precision, os = to.scale;
  i128 maxv = 0;
  pow10_checked(op >= 0 && op <= 38 ? op : 0, &maxv);
  c.p.max_v = maxv - 1;
  c.p.k = 1;
  const int a = os < 0 ? -os : os;
  if (os < 0) {
    i128 k = 0;
    if (!pow10_checked(a, &k) || (u128)k > src_max) c.p.mode = I_ZEROS;  // T::Native 10.pow_checked(..) is None
    else {
      c.p.mode = I_DIV;
      c.p.k = k;
    }
  } else {
    if (!pow10_checked(a, &c.p.k)) {
      c.status = AH_CAST_ERROR;
      c.message = "Cannot cast to \"Decimal128\"(" + std::to_string(op) + ", " + std::to_string(os) + "). The scale causes overflow.";
      return c;
    }
    c.p.mode = I_MUL;
  }
  if (op == 0) {
    c.post_status = AH_INVALID_ARGUMENT;
    c.post_message = "precision cannot be 0, has to be between [1, 38]";
  } else if (op > kMaxPrecision) {
    c.post_status = AH_INVALID_ARGUMENT;
    c.post_message = "precision " + std::to_string(op) + " is greater than max 38";
  } else if (os > kMaxScale) {
    c.post_status = AH_INVALID_ARGUMENT;
    c.post_message = "scale " + std::to_string(os) + " is greater than max 38";
  } else if (os > 0 && os > op) {
    c.post_status = AH_INVALID_ARGUMENT;
    c.post_message = "scale " + std::to_string(os) + " is greater than precision " + std::to_string(op);
  }
  return c;
}
static inline ah_status icast_row_error(const IParams& p, const ah_data_type& to, i128 x, std::string* msg) {
  i128 v = 0;
  int stage = 0;
  icast_row(p, x, &v, &stage);
  if (stage == 1) {  // `.and_then(|v| v.mul_checked(scale_factor))?` keeps mul_checked's own error
    *msg = "Overflow happened on: " + i128_text(x) + " * " + i128_text(p.k);
    return AH_ARITHMETIC_OVERFLOW;
  }
  i128 r = x, rem;
  if (p.mode == I_MUL) da_mul_checked(x, p.k, &r);
  else da_divrem(x, p.k, &r, &rem);
  const bool large = r > p.max_v;
  *msg = format_decimal_str(i128_text(r), (size_t)to.precision, to.scale, false) + (large ? " is too large" : " is too small") +
         " to store in a Decimal128 of precision " + std::to_string(to.precision) + (large ? ". Max is " : ". Min is ") +
         format_decimal_str(i128_text(large ? p.max_v : -p.max_v), (size_t)to.precision, to.scale, true);
  return AH_INVALID_ARGUMENT;
}

}  // namespace da
