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

}  // namespace da
