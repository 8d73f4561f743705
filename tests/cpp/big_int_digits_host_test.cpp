// big_int_digits_host_test.cpp — TEST INFRASTRUCTURE: csrc/big_int_digits.hpp (the PRODUCT header cast_string.hip compiles for
// the GPU) on the CPU against libstdc++'s std::to_chars(double), which prints the shortest round-trip digits (it is Ryu).
// Every integer-valued double in [2^53, 2^64) with a non-zero mantissa field must give the same (digits, power of ten) once
// both are stripped of trailing zeros; exact powers of two and everything outside the range must be declined.
//   usage: big_int_digits_host_test <random cases per exponent> <seed>     exit 0 + "BIG_INT_DIGITS_OK <checked>"
#include <charconv>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../arrow-rs_amd/csrc/big_int_digits.hpp"

static uint64_t st = 0x9E3779B97F4A7C15ull;
static uint64_t rnd() {
  st ^= st >> 12;
  st ^= st << 25;
  st ^= st >> 27;
  return st * 2685821657736338717ull;
}

static long checked = 0, failures = 0;

static void reference(double v, uint64_t* digits, int* exp10) {  // shortest digits d and e with v == d * 10^e, d % 10 != 0
  char buf[64];
  auto r = std::to_chars(buf, buf + sizeof buf, v, std::chars_format::scientific);
  *r.ptr = 0;
  uint64_t d = 0;
  int nd = 0;
  const char* p = buf;
  for (; *p && *p != 'e'; ++p)
    if (*p >= '0' && *p <= '9') d = d * 10 + (uint64_t)(*p - '0'), ++nd;
  const int e = atoi(p + 1);
  int e10 = e - (nd - 1);
  while (d % 10 == 0) d /= 10, ++e10;
  *digits = d, *exp10 = e10;
}

static void check(double v) {
  uint64_t bits;
  memcpy(&bits, &v, 8);
  const uint64_t mant = bits & ((1ull << 52) - 1);
  const uint32_t exp = (uint32_t)(bits >> 52) & 0x7FFu;
  uint64_t m = 0;
  int32_t e = 0;
  const bool took = ah_big_int_shortest(mant, exp, &m, &e);
  const bool in_class = v >= 9007199254740992.0 && v < 18446744073709551616.0 && mant != 0;
  ++checked;
  if (took != in_class) {
    if (++failures < 10) printf("FAIL %.17g: taken %d, in class %d\n", v, (int)took, (int)in_class);
    return;
  }
  if (!took) return;
  while (m % 10 == 0) m /= 10, ++e;
  uint64_t rd;
  int re;
  reference(v, &rd, &re);
  if (m != rd || e != re)
    if (++failures < 10) printf("FAIL %.17g: got %llue%d expected %llue%d\n", v, (unsigned long long)m, e, (unsigned long long)rd, re);
}

int main(int argc, char** argv) {
  const long per_exp = argc > 1 ? atol(argv[1]) : 200000;
  if (argc > 2) st ^= (uint64_t)atoll(argv[2]) * 0xD1B54A32D192ED03ull;
  for (int s = 1; s <= 11; ++s)
    for (long i = 0; i < per_exp; ++i) {
      const uint64_t m2 = (1ull << 52) | (rnd() & ((1ull << 52) - 1));
      check((double)(m2 << s));  // exact: 53 significant bits
    }
  // neighbours of 10^k, 5 * 10^(k-1), 25 * 10^(k-2), 125 * 10^(k-3): long removable runs, ties, "1e19"
  for (int k = 15; k <= 19; ++k) {
    const double bases[4] = {1.0, 0.5, 0.25, 0.125};
    for (double f : bases) {
      double x = f;
      for (int j = 0; j < k; ++j) x *= 10.0;
      double lo = x, hi = x;
      check(x);
      for (int j = 0; j < 2000; ++j) {
        lo = __builtin_nextafter(lo, 0.0), hi = __builtin_nextafter(hi, 1e300);
        check(lo), check(hi);
      }
    }
  }
  // multiples of 10^j and of 5 * 10^(j-1) (ties after j removed digits; inclusive / exclusive ends by mantissa parity)
  for (int j = 1; j <= 8; ++j) {
    uint64_t p = 1;
    for (int q = 0; q < j; ++q) p *= 10;
    for (long i = 0; i < per_exp / 4; ++i) {
      const uint64_t base = ((1ull << 53) / p + rnd() % (((1ull << 63) - (1ull << 53)) / p)) * p;
      check((double)base);
      check((double)(base + p / 2));
    }
  }
  // the class boundaries: powers of two, their neighbours, values just outside
  for (int e = 50; e <= 66; ++e) {
    double x = 1.0;
    for (int j = 0; j < e; ++j) x *= 2.0;
    double lo = x, hi = x;
    check(x);
    for (int j = 0; j < 64; ++j) {
      lo = __builtin_nextafter(lo, 0.0), hi = __builtin_nextafter(hi, 1e300);
      check(lo), check(hi);
    }
  }
  check(0.0), check(1.0), check(1e300), check(123456789.0), check(4503599627370497.0);
  if (failures) {
    printf("BIG_INT_DIGITS_FAILED %ld of %ld\n", failures, checked);
    return 1;
  }
  printf("BIG_INT_DIGITS_OK %ld\n", checked);
  return 0;
}
