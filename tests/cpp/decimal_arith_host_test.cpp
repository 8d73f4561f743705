// Host check of arrow-rs_amd/csrc/decimal_arith.hpp — the SAME header arith_decimal.hip compiles for gfx950.
// `dec_row` (128-bit checked multiply and division written on 64-bit limbs, because the device runtime has neither)
// is compared with the compiler's native __int128 arithmetic (__builtin_*_overflow, `/`, `%`) over edge operands and
// argv[1] random operands of every magnitude.  Prints "ok <cases>".  Test infrastructure only.
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#include "../../arrow-rs_amd/csrc/decimal_arith.hpp"

using namespace da;

static int reference_row(const DParams& p, i128 l, i128 r, i128* out) {
  i128 a = l, b = r;
  if (p.scaled) {
    if (__builtin_mul_overflow(l, p.l_mul, &a)) return D_FAIL_L_SCALE;
    if (__builtin_mul_overflow(r, p.r_mul, &b)) return D_FAIL_R_SCALE;
  }
  switch (p.op) {
    case D_ADD: return __builtin_add_overflow(a, b, out) ? D_FAIL_OP : D_OK;
    case D_SUB: return __builtin_sub_overflow(a, b, out) ? D_FAIL_OP : D_OK;
    case D_MUL: return __builtin_mul_overflow(a, b, out) ? D_FAIL_OP : D_OK;
    default:
      if (b == 0) return D_FAIL_DIV_ZERO;
      if (a == da_min() && b == -1) return D_FAIL_OP;
      *out = p.op == D_DIV ? a / b : a % b;
      return D_OK;
  }
}

int main(int argc, char** argv) {
  const long n = argc > 1 ? atol(argv[1]) : 200000;
  std::mt19937_64 rng(12345);
  const i128 MAXV = (i128)(((u128)1 << 127) - 1), MINV = da_min();
  std::vector<i128> edges = {0, 1, -1, 2, -2, 10, -10, MAXV, MINV, MAXV - 1, MINV + 1, (i128)1 << 64, -((i128)1 << 64),
                             ((i128)1 << 64) - 1, (i128)1 << 63, -((i128)1 << 63), (i128)1 << 126, -((i128)1 << 126),
                             (i128)UINT64_MAX, -(i128)UINT64_MAX, MAXV / 10, MINV / 10, MAXV / 10 + 1, MINV / 10 - 1};
  i128 p10 = 1;
  for (int k = 0; k <= 38; ++k) {
    edges.push_back(p10);
    edges.push_back(-p10);
    edges.push_back(p10 - 1);
    edges.push_back(MAXV / p10);
    edges.push_back(MAXV / p10 + 1);
    edges.push_back(MINV / p10);
    edges.push_back(MINV / p10 - 1);
    if (k < 38) p10 *= 10;
  }
  auto random_value = [&]() -> i128 {
    int bits = (int)(rng() % 128) + 1;  // every magnitude
    u128 v = ((u128)rng() << 64) | rng();
    if (bits < 128) v &= (((u128)1 << bits) - 1);
    return (i128)v * ((rng() & 1) ? 1 : -1);
  };
  long cases = 0, fails = 0;
  auto check = [&](const DParams& p, i128 l, i128 r) {
    i128 got = 0, exp = 0, ls = 0, rs = 0;
    int fg = dec_row(p, l, r, &got, &ls, &rs), fe = reference_row(p, l, r, &exp);
    ++cases;
    if (fg != fe || (fg == D_OK && got != exp)) {
      if (fails++ < 10)
        fprintf(stderr, "mismatch op=%d scaled=%d l=%s r=%s lm=%s rm=%s: got %d %s expected %d %s\n", p.op, p.scaled,
                i128_text(l).c_str(), i128_text(r).c_str(), i128_text(p.l_mul).c_str(), i128_text(p.r_mul).c_str(), fg,
                i128_text(got).c_str(), fe, i128_text(exp).c_str());
    }
  };
  std::vector<i128> muls = {1, 10, 1000, (i128)1000000000000ll, p10 / 100, p10};  // p10 = 10^38
  for (int op = D_ADD; op <= D_REM; ++op)
    for (int scaled = 0; scaled <= 1; ++scaled)
      for (i128 lm : muls)
        for (i128 rm : muls) {
          if (!scaled && (lm != 1 || rm != 1)) continue;
          DParams p{op, scaled, lm, rm};
          for (i128 l : edges)
            for (size_t j = 0; j < edges.size(); j += (scaled ? 7 : 1)) check(p, l, edges[j]);
        }
  for (long i = 0; i < n; ++i) {
    DParams p{(int)(rng() % 5), (int)(rng() & 1), muls[rng() % muls.size()], muls[rng() % muls.size()]};
    check(p, random_value(), random_value());
    // quotients with few bits exercise the short paths of the long division
    i128 d = random_value();
    if (d != 0) check(DParams{D_DIV + (int)(rng() & 1), 0, 1, 1}, d * (i128)(rng() % 1000) + (i128)(rng() % 7), d);
  }
  // dcast_row against a direct statement with native `/`, `%` and __builtin_mul_overflow
  for (int mode = C_UP; mode <= C_DOWN; ++mode)
    for (int delta = 0; delta <= 38; ++delta)
      for (int prec = 1; prec <= 38; prec += 3)
        for (int inf = 0; inf <= 1; ++inf) {
          CParams c{};
          c.mode = mode;
          c.infallible = inf;
          pow10_checked(delta, &c.k);
          if (mode == C_DOWN && delta == 0) continue;
          c.half = c.k / 2;
          i128 mx;
          pow10_checked(prec, &mx);
          c.max_v = mx - 1;
          for (size_t e = 0; e < edges.size() + 40; ++e) {
            i128 x = e < edges.size() ? edges[e] : random_value();
            i128 got = 0, exp = 0;
            int stage = 0;
            bool ok = dcast_row(c, x, &got, &stage), eok = true;
            if (mode == C_UP) {
              if (inf) exp = (i128)((u128)x * (u128)c.k);
              else eok = !__builtin_mul_overflow(x, c.k, &exp);
            } else {
              i128 d = x / c.k, r = x % c.k;
              if (x >= 0 && r >= c.half) d += 1;
              if (x < 0 && r <= -c.half) d -= 1;
              exp = d;
            }
            if (eok && !inf && (exp > c.max_v || exp < -c.max_v)) eok = false;
            ++cases;
            if (ok != eok || (ok && got != exp)) {
              if (fails++ < 10) fprintf(stderr, "cast mismatch mode=%d delta=%d prec=%d inf=%d x=%s\n", mode, delta, prec, inf, i128_text(x).c_str());
            }
          }
        }
  if (format_decimal_str("12345", 5, 2, true) != "123.45" || format_decimal_str("-5", 5, 3, true) != "-0.005" ||
      format_decimal_str("-12345678", 5, 2, true) != "-123.45" || format_decimal_str("12", 5, -2, false) != "1200" ||
      format_decimal_str("-1234", 4, 0, false) != "-1234") {
    fprintf(stderr, "format_decimal_str wrong\n");
    return 1;
  }
  // i128_text against snprintf on the halves
  if (i128_text(MINV) != "-170141183460469231731687303715884105728" || i128_text(MAXV) != "170141183460469231731687303715884105727" ||
      i128_text(0) != "0" || i128_text(-42) != "-42") {
    fprintf(stderr, "i128_text wrong\n");
    return 1;
  }
  if (fails) return 1;
  printf("ok %ld\n", cases);
  return 0;
}
